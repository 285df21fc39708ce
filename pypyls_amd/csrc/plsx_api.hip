// plsx_api.hip -- host side of libplsx.so: context, planning, kernel launches.
// C ABI declared in include/plsx.h.  gfx950 only.
#include "plsx_kernels.h"
#include "plsx_simpls.h"
#include "plsx_resample.h"
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

namespace {

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

// Route / layout switches (plsx_set_option).  Every route computes the same statistics -- they exist for A/B
// measurements and so that the tests can pin each kernel variant against the others and the oracle.  None is
// read from the environment: a host that wants PLSX_<NAME>=1 to mean something translates it itself
// (pypyls_amd.engine.options_from_env, used by bench.py and the tests only).
enum {
    OPT_NO_REFINE = 0, OPT_TRACE_ALLOC, OPT_XPROD_MT24, OPT_MIN_BATCH, OPT_INBLOCK_MOMENTS, OPT_EPI2_NW4,
    OPT_NO_COMPACT_BOOT, OPT_COMPACT_BOOT_ALWAYS, OPT_SEPMOM_ALWAYS, OPT_GRAM_NT, OPT_GRAM_REG, OPT_NO_GRAM4,
    OPT_UROT_NW4, OPT_UROT_GENERIC, OPT_UROT_NO_TAIL4, OPT_NO_FIXED_X, OPT_NO_DUAL_PERM, OPT_TWO_PASS_BOOT,
    OPT_SPLIT_NO_TAIL4, OPT_SPLIT_INBLOCK, OPT_NO_SPLIT_FUSE, OPT_EXPECT_RESAMPLES, OPT_UROT_M3, OPT_SIMPLS_JACOBI, OPT_PERCENTILE_SORT, OPT_QUAD_SUMS, OPT_QUAD_MT, OPT_QUAD_FULL_ROWS, OPT_QUAD_LAUNCH_PER_BLOCK, OPT_COUNT
};
static const char* const kOptionNames[OPT_COUNT] = {
    "no_refine", "trace_alloc", "xprod_mt24", "min_batch", "inblock_moments", "epi2_nw4",
    "no_compact_boot", "compact_boot_always", "sepmom_always", "gram_nt", "gram_reg", "no_gram4",
    "urot_nw4", "urot_generic", "urot_no_tail4", "no_fixed_x", "no_dual_perm", "two_pass_boot",
    "split_no_tail4", "split_inblock", "no_split_fuse", "expect_resamples", "urot_m3", "simpls_jacobi", "percentile_sort", "quad_sums", "quad_mt", "quad_full_rows", "quad_launch_per_block"};

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
    Buf refV, refLam, refK0, refPart, refPartP, refH;                   // graded spectra: parked eigenvectors / eigenvalues / first small rank, partial refined Grams
    int graded = 0;                                     // the ORIGINAL spectrum has live LVs below PLSX_REFINE_TAU d_max: no dual-space routes
    long long n_refined = 0, n_unrefined = 0;           // host copies of status[1], status[2] since the last plsx_numeric_report
    Buf cellS, rowc, out_row_s;                         // fused split-half: cell moments of X, row constants, row map
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
    long long R_geom[3] = {0, 0, 0};                    // (T', T'pp, Bpad) the R scratch was last zeroed under
    size_t R_zeroed_bytes = 0;
    double map_ms_per_gb = 0.0;                         // measured cost of mapping device memory (launch_groups), 0 = not yet
    int scratch_fixed = 0;                              // 1: always launch budget-sized super-batches
                          // resamples covered by the timed launches
    double last_ms = 0.0;
    int last_launches = 0;
};

namespace {

int fail(plsx_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

// Nothing may unwind through the C ABI (include/plsx.h): every extern "C" entry with a body worth guarding is a
// function-try-block closed by PLSX_CATCH -- std::vector / std::string / std::thread inside the library can throw
// (host memory, thread limits), and so could anything added later.  The handlers themselves do not throw.
int fail_nothrow(plsx_ctx* c, int code, const char* what) noexcept
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

int ensure(plsx_ctx* ctx, Buf& b, size_t bytes, bool zero = false)
{
    if (b.bytes < bytes) {
        const bool trace = ctx->opt[OPT_TRACE_ALLOC] != 0;
        auto t0 = std::chrono::steady_clock::now();
        if (b.p) HIPCHK(hipFree(b.p));
        auto t1 = std::chrono::steady_clock::now();
        b.p = nullptr;
        b.bytes = 0;
        HIPCHK(hipMalloc(&b.p, bytes));
        auto t2 = std::chrono::steady_clock::now();
        b.bytes = bytes;
        if (zero) HIPCHK(hipMemset(b.p, 0, bytes));
        if (trace) {
            HIPCHK(hipDeviceSynchronize());
            auto t3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
            fprintf(stderr, "[plsx alloc] %.3f GB: free %.1f ms, malloc %.1f ms, zero %.1f ms\n",
                    bytes / 1073741824.0, ms(t0, t1), ms(t1, t2), ms(t2, t3));
        }
    }
    return 0;
}

void release(Buf& b)
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
const char* const kKernelClassNames[KC_COUNT] = {"k_xprod", "k_gram", "k_small", "k_urot", "k_nt_gemm",
                                                 "k_ucorr_partial", "k_simpls_dual", "k_build_A", "k_xprod_moments"};

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

int ceil_div(int a, int b) { return (a + b - 1) / b; }
int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Physical cross-product groups of `rgroups` resample groups.
int phys_groups(const plsx_ctx* c, int rgroups) { return c->gps > 0 ? rgroups * c->gps : rgroups; }

// Choose resamples per group so that data + moment tiles fill MT tiles; when one
// resample does not fit a block, cut its rows into slices (one group each).
int plan_groups(plsx_ctx* c)
{
    c->scaled = (c->method == PLSX_BEHAVIORAL && !c->cov) ? 1 : 0;   // mean-centred / regression: no feature scaling
    // covariance mode scales nothing: its moment rows only serve cross-validation's zmap and are
    // planned in for the duration of plsx_crossval_batch (cv_mom), not for permutations / bootstraps
    c->momrows = (c->method == PLSX_BEHAVIORAL && (!c->cov || c->cv_mom)) ? 1 : 0;
    const int Jw = c->momrows ? c->J : 0;
    c->gps = 0;
    c->h_slice_row0.clear(); c->h_slice_rows.clear(); c->h_slice_cell0.clear(); c->h_slice_ncell.clear();
    auto fit = [&](int mt) {
        int b = 0;
        for (int n = 1; n <= 512; ++n) {
            int td = ceil_div(n * c->Tp, 16), tw = Jw ? ceil_div(n * Jw, 16) : 0;
            if (td + 2 * tw <= mt && tw * 16 <= 48) b = n; else break;
        }
        return b;
    };
    c->MT = 24;
    int best = fit(24);
    // a block of 16 tiles when it wastes clearly fewer rows (one resample of 177 <= T' <= 224 rows
    // fills 15 of 24 tiles but 15 of 16); behavioural correlation PLS only (instantiations of k_xprod)
    if (c->method == PLSX_BEHAVIORAL && !c->opt[OPT_XPROD_MT24]) {
        const int b16 = fit(16);
        if (b16 >= 1 && (double)b16 / 16.0 > 1.15 * (double)best / 24.0) { c->MT = 16; best = b16; }
    }
    int tw = Jw ? ceil_div(std::max(best, 1) * Jw, 16) : 0;
    if (best == 0) {
        // sliced: greedy row ranges [a, b) with their cells' moment rows
        const int Tc = (c->method == PLSX_BEHAVIORAL) ? c->T : 1;      // rows per cell
        int a = 0, twmax = 0;
        while (a < c->Tp) {
            int bsel = -1, twsel = 0;
            for (int dt = c->MT; dt >= 1 && bsel < 0; --dt) {
                const int b = std::min(c->Tp, a + dt * 16);
                const int nc = Jw ? ((b - 1) / Tc - a / Tc + 1) : 0;
                const int t2 = ceil_div(nc, 16);
                if (ceil_div(b - a, 16) + 2 * t2 <= c->MT && t2 <= 3) { bsel = b; twsel = t2; }
            }
            if (bsel < 0) return -1;
            c->h_slice_row0.push_back(a);
            c->h_slice_rows.push_back(bsel - a);
            c->h_slice_cell0.push_back(Jw ? a / Tc : 0);
            c->h_slice_ncell.push_back(Jw ? ((bsel - 1) / Tc - a / Tc + 1) : 0);
            twmax = std::max(twmax, twsel);
            a = bsel;
        }
        c->gps = (int)c->h_slice_row0.size();
        best = 1;
        tw = twmax;
    }
    c->npg = best;
    // tile order inside a group: data tiles, (unused tiles,) first-moment
    // (weight) tiles, second-moment tiles LAST (the kernel's static split)
    c->sq0 = c->MT - tw;
    c->w0 = c->sq0 - tw;
    c->nmom_pad = tw * 16;
    c->group_stride = (size_t)c->nks * c->MT * 64;
    const int ncolblk = c->Bpad / 128;
    // super-batch = g groups.  Large enough that (a) the cross-product grid
    // covers the chip many times over and (b) the latency-bound small-solver
    // launch (one block per resample) has >= 2 blocks per CU to overlap.
    int g = round_up(std::max(1, ceil_div(2048, ncolblk)), 8);
    // ... and (c) small shapes amortise their launches: 4096 resamples per super-batch where the budget
    // allows (c2: 512 -> 4096 per batch, 1.72 M -> 1.91 M resamples/s; the headline shape is budget bound)
    g = std::max(g, round_up(ceil_div(c->opt[OPT_MIN_BATCH] > 0 ? c->opt[OPT_MIN_BATCH] : 4096, std::max(best, 1)), 8));
    g = std::min(std::max(g, 8), 128);
    const double budget = c->scratch_gb * 1073741824.0;
    while (g > 1 && (double)g * best * c->Tpp * (double)c->Bpad * 8.0 > budget) g -= (g > 8 ? 8 : 1);
    c->Gcap = g;
    return 0;
}

int upload_rowmaps(plsx_ctx* ctx)
{
    const int rows = ctx->MT * 16, ntab = std::max(ctx->gps, 1);
    std::vector<int> out_row((size_t)ntab * rows, -1), mom_idx((size_t)ntab * rows, -1);
    if (ctx->gps > 0) {
        const int Tc = (ctx->method == PLSX_BEHAVIORAL) ? ctx->T : 1;
        std::vector<int> row_slice(ctx->Tp), row_local(ctx->Tp), cell_momrow(std::max(ctx->J, 1), 0);
        for (int sl = 0; sl < ctx->gps; ++sl)
            for (int k = 0; k < ctx->h_slice_rows[sl]; ++k) {
                const int grow = ctx->h_slice_row0[sl] + k;
                row_slice[grow] = sl;
                row_local[grow] = k;
                out_row[(size_t)sl * rows + k] = grow;
                if (ctx->scaled) mom_idx[(size_t)sl * rows + k] = grow / Tc - ctx->h_slice_cell0[sl];
            }
        if (ctx->momrows)
            for (int j = 0; j < ctx->J; ++j) {
                const int s0 = row_slice[j * Tc];
                cell_momrow[j] = s0 * ctx->nmom_pad + (j - ctx->h_slice_cell0[s0]);
            }
        if (ensure(ctx, ctx->row_slice, ctx->Tp * sizeof(int))) return PLSX_ERR_HIP;
        if (ensure(ctx, ctx->row_local, ctx->Tp * sizeof(int))) return PLSX_ERR_HIP;
        if (ensure(ctx, ctx->slice_cell0, ctx->gps * sizeof(int))) return PLSX_ERR_HIP;
        HIPCHK(hipMemcpy(ctx->row_slice.p, row_slice.data(), ctx->Tp * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->row_local.p, row_local.data(), ctx->Tp * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->slice_cell0.p, ctx->h_slice_cell0.data(), ctx->gps * sizeof(int), hipMemcpyHostToDevice));
        if (ensure(ctx, ctx->cell_momrow, cell_momrow.size() * sizeof(int))) return PLSX_ERR_HIP;
        HIPCHK(hipMemcpy(ctx->cell_momrow.p, cell_momrow.data(), cell_momrow.size() * sizeof(int), hipMemcpyHostToDevice));
    } else {
        for (int rr = 0; rr < ctx->npg; ++rr)
            for (int t = 0; t < ctx->Tp; ++t) {
                int row = rr * ctx->Tp + t;
                out_row[row] = rr * ctx->Tpp + t;
                if (ctx->scaled) mom_idx[row] = rr * ctx->J + t / ctx->T;
            }
    }
    if (ensure(ctx, ctx->out_row, out_row.size() * sizeof(int))) return PLSX_ERR_HIP;
    if (ensure(ctx, ctx->mom_idx, mom_idx.size() * sizeof(int))) return PLSX_ERR_HIP;
    HIPCHK(hipMemcpy(ctx->out_row.p, out_row.data(), out_row.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->mom_idx.p, mom_idx.data(), mom_idx.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

// Separate-moments layout (correlation mode, plain layout): choose the data block height and
// upload its row maps.  In-block moments cost 2 of 24 tiles for 7 + 7 rows at the headline shape.
int plan_sepmom(plsx_ctx* ctx)
{
    ctx->sepmom = 0;
    if (!ctx->scaled || ctx->gps > 0 || ctx->opt[OPT_INBLOCK_MOMENTS]) return 0;
    int best_mt = 0, best_n = 0;
    double best_fill = 0.0;
    for (int mt : {24, 22, 16}) {
        const int n = (mt * 16) / ctx->Tp;
        if (n < 1) continue;
        // rows used per tile; a lower block re-reads X more often: it has to win by 3 %
        const double fill = (double)n * ctx->Tp / (mt * 16.0) * (mt == 24 ? 1.0 : (mt == 22 ? 0.985 : 0.955));
        if (fill > best_fill) { best_fill = fill; best_mt = mt; best_n = n; }
    }
    if (best_n < ctx->npg || best_n * ctx->J * 64 * 8 > 48 * 1024) return 0;     // scale tile of a block in LDS
    ctx->MTd = best_mt; ctx->npg_d = best_n;
    ctx->group_stride_d = (size_t)ctx->nks * best_mt * 64;
    const int rows = best_mt * 16;
    std::vector<int> orow(rows, -1), mrow(rows, -1);
    for (int rr = 0; rr < best_n; ++rr)
        for (int t = 0; t < ctx->Tp; ++t) {
            orow[rr * ctx->Tp + t] = rr * ctx->Tpp + t;
            mrow[rr * ctx->Tp + t] = rr * ctx->J + t / ctx->T;
        }
    if (ensure(ctx, ctx->out_row_d, rows * sizeof(int))) return PLSX_ERR_HIP;
    if (ensure(ctx, ctx->mom_idx_d, rows * sizeof(int))) return PLSX_ERR_HIP;
    HIPCHK(hipMemcpy(ctx->out_row_d.p, orow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->mom_idx_d.p, mrow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    ctx->sepmom = 1;
    return 0;
}

// scratch for `groups` groups of resamples
int ensure_scratch(plsx_ctx* ctx, int groups)
{
    groups = std::min(std::max(groups, 1), ctx->Gcap);
    if (groups <= ctx->Galloc) return 0;
    // resamples held at once: the fixed-X path packs more resamples per group
    const size_t nb = (size_t)groups * std::max(ctx->npg, ctx->npgf);
    const size_t astride = std::max(ctx->group_stride, ctx->group_stride_f);
    const size_t pg = (size_t)phys_groups(ctx, groups);
    if (int e = ensure(ctx, ctx->Afrag, pg * astride * 8 + 4096)) return e;
    // R: rows t >= Tp of every resample stay zero forever (memset on alloc)
    if (int e = ensure(ctx, ctx->R, nb * ctx->Tpp * (size_t)ctx->Bpad * 8, true)) return e;
    if (int e = ensure(ctx, ctx->mom_n, pg * std::max(ctx->nmom_pad, 16) * 8, true)) return e;
    if (int e = ensure(ctx, ctx->Gm, nb * ctx->Tp * ctx->Tp * 8)) return e;
    if (int e = ensure(ctx, ctx->Pm, nb * ctx->Tp * ctx->L * 8)) return e;
    if (int e = ensure(ctx, ctx->Mfrag, nb * ctx->nks_t * ctx->LT * 64 * 8 + 1024)) return e;   // + one DMA piece of slack
    ctx->Galloc = groups;
    return 0;
}

// Groups per launch for a call that processes `units` resamples packed
// `per_group` to a group.  A fixed budget (plsx_set_scratch / PLSX_SCRATCH_GB)
// always launches budget-sized super-batches: best steady-state throughput for
// a long-lived context.  Otherwise the size weighs the cost of mapping device
// memory against the fixed cost per launch (~2.5 ms: one wave of the
// latency-bound small solver plus fills).  Mapping is not free on a shared
// MI355X: the driver clears recycled VRAM lazily, and a request that outgrows
// the pool of already-clean pages (a few tens of GB) stalls for ~25 ms per GB
// of dirty memory on the device -- seconds, measured 2-6 s -- which a one-shot
// call of a few thousand resamples should not pay to compute for one second.
// The model prices that at 40 ms per GB requested.  Scratch that is already
// mapped is always used in full.
int launch_groups(plsx_ctx* ctx, long long units, int per_group)
{
    // a caller that ships one analysis in chunks (as the index rows arrive) says how many resamples are coming
    // ("expect_resamples"): the scratch is then sized once, for the whole shard, instead of growing chunk by
    // chunk -- every growth is a hipFree (a device-wide sync in the middle of the queue) + hipMalloc + zero fill
    units = std::max<long long>(units, ctx->opt[OPT_EXPECT_RESAMPLES]);
    const long long need = (units + per_group - 1) / std::max(per_group, 1);
    const int cap = (int)std::max<long long>(1, std::min<long long>(ctx->Gcap, need));
    if (ctx->scratch_fixed) return cap;
    const double gb_per_group = (double)std::max(ctx->npg, ctx->npgf) * ctx->Tpp * (double)ctx->Bpad * 8.0 /
                                1073741824.0;
    // per-launch cost: one wave of the small solver -- 2.5 ms for the LDS Jacobi variant, ~25 ms
    // x (T'/200)^3 for Householder + QL (T' > PLSX_JACOBI_TP; one block per resample, latency
    // bound: only a large batch keeps the chip busy)
    const double tn = ctx->Tp / 200.0;
    // what mapping a GB costs HERE is measured once per context (2 GB: hipMalloc + first touch + free) instead of
    // assumed: ~1 ms per GB on a device with clean pages (the model then launches 2 - 3 x larger super-batches:
    // a 1250-bootstrap shard of c4 ran 56 per launch under the fixed 40 ms per GB and paid one 2.5 ms wave of the
    // small solver per 10 ms of cross-product), 25+ ms per GB when the driver has to clear recycled VRAM first.
    // Floor 1.5 ms per GB (the zero fill of R and a margin for the pool running dry beyond the probe), cap 40.
    if (ctx->map_ms_per_gb <= 0.0) {
        ctx->map_ms_per_gb = 40.0;
        void* probe = nullptr;
        const size_t pb = (size_t)2 << 30;
        auto t0 = std::chrono::steady_clock::now();
        if (hipMalloc(&probe, pb) == hipSuccess) {
            (void)hipMemset(probe, 0, pb);
            (void)hipDeviceSynchronize();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            (void)hipFree(probe);
            ctx->map_ms_per_gb = std::min(40.0, std::max(1.5, 2.0 * (ms / 2.0)));   // (safety factor 2 on the per-GB time)
        } else (void)hipGetLastError();
    }
    const double c_group = ctx->map_ms_per_gb * gb_per_group, c_launch = ctx->Tp > PLSX_JACOBI_TP ? std::max(2.5, 25.0 * tn * tn * tn) : 2.5;
    int g = round_up((int)std::ceil(std::sqrt(c_launch * (double)need / std::max(c_group, 1e-3))), 8);
    g = std::max(g, ctx->Galloc);
    // a re-bound context (the front-ends' cached engine) keeps the R it mapped for an earlier call: groups that fit
    // what is already there cost nothing to map
    if (ctx->R.bytes > 0 && gb_per_group > 0.0)
        g = std::max(g, (int)std::min<double>(cap, std::floor((double)ctx->R.bytes / (gb_per_group * 1073741824.0))));
    return std::max(1, std::min(g, cap));
}

// Super-batches of a call of n resamples with at most `cap` per launch (cap a multiple of `per_group`): the same
// number of launches, but of equal size -- 625 bootstraps run as 315 + 310, not 504 + 121 (a launch of 121 costs
// the latency-bound stages, one wave of the small solver and the moment blocks, as much as one of 504).
int balanced_batch(int n, int cap, int per_group)
{
    const int launches = ceil_div(n, std::max(cap, 1));
    return std::min(cap, round_up(ceil_div(n, launches), std::max(per_group, 1)));
}

template <int MT, int NW, int KT, int NSQ>
int launch_xprod_t(plsx_ctx* ctx, int groups, hipStream_t st)
{
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    const size_t epi = (size_t)NW * 2 * NSQ * 16 * 16 * 8 + (size_t)2 * MT * 16 * 4;
    const size_t lds = std::max(stage, epi);
    HIPCHK(set_lds(k_xprod<MT, NW, KT, NSQ>, lds));
    const int ncolblk = ctx->Bpad / (NW * 16);
    dim3 grid(ncolblk * round_up(groups, 8)), block(NW * 64);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod<MT, NW, KT, NSQ>), grid, block, lds, st,
                       ptr<double>(ctx->Afrag) + (size_t)ctx->afrag_group0 * ctx->group_stride, ctx->group_stride,
                       ptr<double>(ctx->Xc), ctx->Bpad,
                       ctx->nks, ptr<double>(ctx->R), ctx->Bpad, ctx->npg * ctx->Tpp,
                       ptr<int>(ctx->out_row), ptr<int>(ctx->mom_idx), ptr<double>(ctx->mom_n),
                       std::max(ctx->nmom_pad, 0), groups, ncolblk, ctx->mom_out_arg,
                       SplitEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}, std::max(ctx->gps, 1));
    LAUNCHCHK();
    return 0;
}

// `groups` = physical groups (phys_groups of the resample groups)
int launch_xprod(plsx_ctx* ctx, int groups, hipStream_t st)
{
    if (ctx->MT == 16)
        switch (ctx->MT - ctx->sq0) {
            case 0: return launch_xprod_t<16, 4, 1, 0>(ctx, groups, st);
            case 1: return launch_xprod_t<16, 4, 1, 1>(ctx, groups, st);
            case 2: return launch_xprod_t<16, 4, 1, 2>(ctx, groups, st);
            default: return launch_xprod_t<16, 4, 1, 3>(ctx, groups, st);
        }
    switch (ctx->MT - ctx->sq0) {          // number of second-moment tiles (0..3)
        case 0: return launch_xprod_t<24, 4, 1, 0>(ctx, groups, st);
        case 1: return launch_xprod_t<24, 4, 1, 1>(ctx, groups, st);
        case 2: return launch_xprod_t<24, 4, 1, 2>(ctx, groups, st);
        default: return launch_xprod_t<24, 4, 1, 3>(ctx, groups, st);
    }
}

// Single-pass bootstraps: the cross-product with an accumulating epilogue (k_xprod EPI 2; A = W_r^T of the
// group's resamples, 24 tiles, row -> LV map `out_row_w`).  Blocks of 8 waves = 128 feature columns: every block
// streams the group's whole A operand through LDS, so twice the columns per block is half the A traffic per
// flop -- at S = 1000 (A = 3 MB per group against the XCD's 4 MB L2, which the X stream keeps evicting) the
// 64-column blocks re-fetched 20 % of A from HBM: c5 49.8 -> 45.8 ms.  PLSX_EPI2_NW4 keeps the 4-wave blocks.
int launch_xprod_acc(plsx_ctx* ctx, const double* Afrag, size_t gstride, int groups, int L, hipStream_t st)
{
    constexpr int MT = 24, KT = 1;
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    SplitEpi se;
    memset(&se, 0, sizeof(se));
    se.acc_sum = ptr<double>(ctx->psum); se.acc_sq = ptr<double>(ctx->psq); se.accL = L; se.accB = ctx->B;
    const bool narrow = ctx->opt[OPT_EPI2_NW4] != 0;
    KTimer tm(ctx, KC_XPROD, st);
    if (!narrow && 2 * (size_t)L * (8 * 16 + 16) * 8 + MT * 16 * 4 <= 96 * 1024) {
        constexpr int NW = 8;
        const size_t lds = std::max(stage, (size_t)2 * L * (NW * 16 + 16) * 8 + (size_t)MT * 16 * 4);
        HIPCHK(set_lds(k_xprod<MT, NW, KT, 0, 2>, lds));
        const int ncolblk = ceil_div(ctx->Bpad, NW * 16);
        hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0, 2>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), lds, st,
                           Afrag, gstride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks, (double*)nullptr, ctx->Bpad, 0,
                           ptr<int>(ctx->out_row_w), (const int*)nullptr, (const double*)nullptr, 0, groups, ncolblk,
                           (double*)nullptr, se, 1);
    } else {
        constexpr int NW = 4;
        const size_t lds = std::max(stage, (size_t)2 * L * PLSX_ACC_PITCH * 8 + (size_t)MT * 16 * 4);
        HIPCHK(set_lds(k_xprod<MT, NW, KT, 0, 2>, lds));
        const int ncolblk = ctx->Bpad / (NW * 16);
        hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0, 2>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), lds, st,
                           Afrag, gstride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks, (double*)nullptr, ctx->Bpad, 0,
                           ptr<int>(ctx->out_row_w), (const int*)nullptr, (const double*)nullptr, 0, groups, ncolblk,
                           (double*)nullptr, se, 1);
    }
    LAUNCHCHK();
    return 0;
}

// Fixed-X fast path: A = z-scored (permuted) Y only, X pre-scaled per cell, no
// moment tiles, 25 M-tiles = 8 resamples of T' = 50 with no padding.
int run_xprod_fixed(plsx_ctx* ctx, const int* ysrc, int nres, hipStream_t st, const double* ystack)
{
    const int groups = ceil_div(nres, ctx->npgf);
    if (int e = ensure_scratch(ctx, groups)) return e;
    if (ctx->timing) ctx->timed_units += nres;
    HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)groups * ctx->group_stride_f * 8, st));
    GroupLayout lay;
    lay.n = ctx->npgf; lay.Tp = ctx->Tp; lay.J = ctx->J; lay.T = ctx->T; lay.MT = ctx->MTf;
    lay.w0 = ctx->MTf; lay.sq0 = ctx->MTf; lay.Tpp = ctx->Tpp;
    {
        dim3 grid(nres, ctx->J), block(256);
        const size_t lds = (size_t)2 * ctx->T * 8;
        hipLaunchKernelGGL(k_build_A_behav, grid, block, lds, st, ystack ? ystack : ptr<double>(ctx->Y),
                           ystack ? (long long)ctx->S * ctx->T : 0LL, ctx->T, ctx->S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), (const int*)nullptr, ysrc, lay,
                           0, 0, ptr<double>(ctx->Afrag), ctx->group_stride_f, ptr<double>(ctx->mom_n), 16);
        LAUNCHCHK();
    }
    constexpr int MT = 25, NW = 4, KT = 1;              // (8-wave blocks measured: no gain here, A = 1.6 MB stays in L2)
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    const size_t lds = std::max(stage, (size_t)2 * MT * 16 * 4);
    HIPCHK(set_lds(k_xprod<MT, NW, KT, 0>, lds));
    const int ncolblk = ctx->Bpad / (NW * 16);
    dim3 grid(ncolblk * round_up(groups, 8)), block(NW * 64);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0>), grid, block, lds, st,
                       ptr<double>(ctx->Afrag), ctx->group_stride_f, ptr<double>(ctx->Xn), ctx->Bpad,
                       ctx->nks, ptr<double>(ctx->R), ctx->Bpad, ctx->npgf * ctx->Tpp,
                       ptr<int>(ctx->out_row_f), ptr<int>(ctx->mom_idx_f), ptr<double>(ctx->mom_n), 0,
                       groups, ncolblk, (double*)nullptr, SplitEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}, 1);
    LAUNCHCHK();
    return 0;
}

// Correlation mode, separate-moments layout: data-only blocks (MTd tiles, npg_d resamples) scaled by
// 1 / std from a table that moment-only blocks (192 (resample, cell) pairs each) write first.
template <int MT>
int launch_xprod_data(plsx_ctx* ctx, int groups, SplitEpi se, hipStream_t st)
{
    constexpr int NW = 4, KT = 1;
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    const size_t epi = (size_t)se.npairs * NW * 16 * 8 + (size_t)2 * MT * 16 * 4;
    const size_t lds = std::max(stage, epi);
    HIPCHK(set_lds(k_xprod<MT, NW, KT, 0, 3>, lds));
    const int ncolblk = ctx->Bpad / (NW * 16);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0, 3>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), lds, st,
                       ptr<double>(ctx->Afrag), ctx->group_stride_d, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                       ptr<double>(ctx->R), ctx->Bpad, ctx->npg_d * ctx->Tpp, ptr<int>(ctx->out_row_d),
                       ptr<int>(ctx->mom_idx_d), (const double*)nullptr, 0, groups, ncolblk, (double*)nullptr, se, 1);
    LAUNCHCHK();
    return 0;
}

int run_xprod_sepmom(plsx_ctx* ctx, const int* xsrc, const int* ysrc, int nres, hipStream_t st,
                     const double* ystack, long long ystride)
{
    const int npairs = nres * ctx->J;
    const int groups_d = ceil_div(nres, ctx->npg_d), groups_m = ceil_div(npairs, PLSX_MOM_PAIRS);
    const size_t mstride = (size_t)ctx->nks * 24 * 64;
    if (int e = ensure(ctx, ctx->Afrag_m, (size_t)groups_m * mstride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->momn_m, (size_t)round_up(npairs, PLSX_MOM_PAIRS) * 8)) return e;
    if (int e = ensure(ctx, ctx->scale, (size_t)round_up(std::max(npairs, groups_d * ctx->npg_d * ctx->J), 8) * ctx->Bpad * 8))
        return e;
    HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)groups_d * ctx->group_stride_d * 8, st));
    HIPCHK(hipMemsetAsync(ctx->Afrag_m.p, 0, (size_t)groups_m * mstride * 8, st));
    GroupLayout lay;
    lay.n = ctx->npg_d; lay.Tp = ctx->Tp; lay.J = ctx->J; lay.T = ctx->T; lay.MT = ctx->MTd;
    lay.w0 = ctx->MTd; lay.sq0 = ctx->MTd; lay.Tpp = ctx->Tpp;
    {
        KTimer tm(ctx, KC_BUILD, st);
        hipLaunchKernelGGL(k_build_A_behav, dim3(nres, ctx->J), dim3(256), (size_t)2 * ctx->T * 8, st,
                           ystack ? ystack : ptr<double>(ctx->Y), ystack ? ystride : 0LL, ctx->T, ctx->S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), xsrc, ysrc, lay, ctx->cov, 1,
                           ptr<double>(ctx->Afrag), ctx->group_stride_d, ptr<double>(ctx->momn_m), 0, 0,
                           ptr<double>(ctx->Afrag_m), mstride);
        LAUNCHCHK();
    }
    SplitEpi se;
    memset(&se, 0, sizeof(se));
    se.scale = ptr<double>(ctx->scale);
    {
        // moment-only blocks: 12 weight tiles against X, the same 12 against X^2
        constexpr int NW = 8;     // (moment-only blocks of 8 waves: half the A traffic per flop)
        const size_t lds = (size_t)2 * (((size_t)24 * 64 + 127) / 128) * 128 * 8;
        HIPCHK(set_lds(k_xprod<24, NW, 1, 12, 4>, lds));
        const int ncolblk = ceil_div(ctx->Bpad, NW * 16);
        se.npairs = npairs;
        KTimer tm(ctx, KC_MOM, st);
        hipLaunchKernelGGL((k_xprod<24, NW, 1, 12, 4>), dim3(ncolblk * round_up(groups_m, 8)), dim3(NW * 64), lds, st,
                           ptr<double>(ctx->Afrag_m), mstride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                           (double*)nullptr, ctx->Bpad, 0, (const int*)nullptr, (const int*)nullptr,
                           ptr<double>(ctx->momn_m), 0, groups_m, ncolblk, (double*)nullptr, se, 1);
        LAUNCHCHK();
    }
    se.npairs = ctx->npg_d * ctx->J;
    se.accB = nres * ctx->Tpp;
    switch (ctx->MTd) {
        case 24: return launch_xprod_data<24>(ctx, groups_d, se, st);
        case 22: return launch_xprod_data<22>(ctx, groups_d, se, st);
        default: return launch_xprod_data<16>(ctx, groups_d, se, st);
    }
}

// Moment-only blocks of a launch of `npairs` (resample, cell) pairs: 192 pairs per block (12 + 12 tiles) or 128
// (8 + 8), whichever issues fewer tiles -- 100 pairs (a split-half pass): 16 instead of 24, 504: 64 instead of 72.
struct MomLayout { int pairs, mt, groups; size_t stride; };
MomLayout moment_layout(const plsx_ctx* ctx, int npairs)
{
    const int t192 = ceil_div(npairs, 192) * 24, t128 = ceil_div(npairs, 128) * 16;
    MomLayout m;
    m.pairs = (t128 * 6 < t192 * 5) ? 128 : 192;        // (a 16-tile block is ~10 % slower per tile: 504 pairs, 64 vs 72 tiles, measured no gain)
    m.mt = m.pairs / 8;
    m.groups = ceil_div(npairs, m.pairs);
    m.stride = (size_t)ctx->nks * m.mt * 64;
    return m;
}

// EPI 4 (1 / std table) or EPI 6 (raw m1, m2) over the layout above; 8-wave blocks for the single table.
template <int EPI>
int launch_moment_blocks(plsx_ctx* ctx, const MomLayout& ml, SplitEpi se, hipStream_t st)
{
    constexpr int NW = (EPI == 4) ? 8 : 4;
    const int ncolblk = ceil_div(ctx->Bpad, NW * 16);
    KTimer tm(ctx, KC_MOM, st);
    if (ml.mt == 24) {
        const size_t lds = (size_t)2 * (((size_t)24 * 64 + 127) / 128) * 128 * 8;
        HIPCHK(set_lds(k_xprod<24, NW, 1, 12, EPI>, lds));
        hipLaunchKernelGGL((k_xprod<24, NW, 1, 12, EPI>), dim3(ncolblk * round_up(ml.groups, 8)), dim3(NW * 64), lds, st,
                           ptr<double>(ctx->Afrag_m), ml.stride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                           (double*)nullptr, ctx->Bpad, 0, (const int*)nullptr, (const int*)nullptr,
                           ptr<double>(ctx->momn_m), 0, ml.groups, ncolblk, (double*)nullptr, se, 1);
    } else {
        const size_t lds = (size_t)2 * (((size_t)16 * 64 + 127) / 128) * 128 * 8;
        HIPCHK(set_lds(k_xprod<16, NW, 1, 8, EPI>, lds));
        hipLaunchKernelGGL((k_xprod<16, NW, 1, 8, EPI>), dim3(ncolblk * round_up(ml.groups, 8)), dim3(NW * 64), lds, st,
                           ptr<double>(ctx->Afrag_m), ml.stride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                           (double*)nullptr, ctx->Bpad, 0, (const int*)nullptr, (const int*)nullptr,
                           ptr<double>(ctx->momn_m), 0, ml.groups, ncolblk, (double*)nullptr, se, 1);
    }
    LAUNCHCHK();
    return 0;
}

// Row maps of a compact block (one resample / split per group): data row t -> R row t, moment index = its cell.
int ensure_compact_maps(plsx_ctx* ctx)
{
    if (ctx->has_compact_maps) return 0;
    const int rows = ceil_div(ctx->Tp, 16) * 16;
    std::vector<int> orow(rows, -1), mrow(rows, -1);
    for (int t = 0; t < ctx->Tp; ++t) { orow[t] = t; mrow[t] = t / ctx->T; }
    if (int e = ensure(ctx, ctx->out_row_c, rows * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->mom_idx_c, rows * sizeof(int))) return e;
    HIPCHK(hipMemcpy(ctx->out_row_c.p, orow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->mom_idx_c.p, mrow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    ctx->has_compact_maps = 1;
    return 0;
}

// Compact bootstraps (correlation mode, T' <= 208): a bootstrap draws ~63 % of the rows of X; the 7-per-block
// layout contracts every block over all S rows (the union of seven draws), i.e. multiplies 37 % zeros.
// Here every bootstrap has a block of its own that contracts over the rows it draws (k_xprod IDX: row table,
// multiplicities folded into A), scaled by the 1 / std table of the moment-only blocks as in the
// separate-moments layout.  ceil(T'/16) tiles x ~0.632 S/4 k-steps instead of 24 tiles x S/4 k-steps per 7.
template <int MT, int KT, bool TAIL = false>
int launch_xprod_cboot(plsx_ctx* ctx, int nres, int nks_c, SplitEpi se, hipStream_t st)
{
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8 + (size_t)nks_c * 4 * sizeof(int);
    const size_t epi = (size_t)se.npairs * 128 * 8 + (size_t)2 * MT * 16 * 4;
    const size_t lds = std::max(stage, epi);
    HIPCHK(set_lds(k_xprod_compact<MT, KT, 3, TAIL>, lds));
    const int ncolblk = ceil_div(ctx->Bpad, 128);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod_compact<MT, KT, 3, TAIL>), dim3(round_up(ncolblk, 8) * round_up(nres, 8)), dim3(256),
                       lds, st, ptr<double>(ctx->Afrag_c), (size_t)nks_c * MT * 64, ptr<double>(ctx->Xc), ctx->Bpad, nks_c,
                       ptr<double>(ctx->R), ctx->Bpad, ctx->Tpp, ptr<int>(ctx->out_row_c), ptr<int>(ctx->mom_idx_c),
                       (const double*)nullptr, nres, ncolblk, se);
    LAUNCHCHK();
    return 0;
}

bool compact_boot_ok(const plsx_ctx* ctx)
{
    // (LDS of a block: the row table, 4 S bytes, behind the A stages; the scale tile of its cells, 1 KB each)
    if (!(ctx->scaled && ctx->method == PLSX_BEHAVIORAL && ctx->gps == 0 && ctx->Tp <= 208 && !ctx->mom_out_arg &&
          ctx->J <= 32 && ctx->S <= 8192 && (long long)ctx->Kpad * ctx->Bpad * 8 < (1LL << 31)))
        return false;
    const int force = ctx->opt[OPT_COMPACT_BOOT_ALWAYS] ? 1 : (ctx->opt[OPT_NO_COMPACT_BOOT] ? -1 : 0);
    if (force) return force > 0;
    // matrix-pipe cycles per bootstrap in units of (S / 4 k-steps x one 16-row tile): a compact block contracts
    // ~0.66 S rows (distinct draws, rounded to k-steps) on its own tiles at ~0.8 of the dense blocks' pipe
    // utilisation (measured at the headline shape); a dense block shares its tiles between its resamples.
    // A dense block that packs many resamples (small T') also re-reads X that much less often: keep it.
    const int mt = ceil_div(ctx->Tp, 16);
    const double rows = (mt >= 2 && ctx->Tp - (mt - 1) * 16 <= 4) ? (mt - 1) * 16 + 4 : mt * 16;
    const double cost_c = 0.66 * 1.25 * rows / 16.0;
    const int npg_dense = ctx->sepmom ? ctx->npg_d : ctx->npg, mt_dense = ctx->sepmom ? ctx->MTd : ctx->MT;
    return cost_c * 1.1 < (double)mt_dense / npg_dense && npg_dense <= 16;
}

int run_xprod_cboot(plsx_ctx* ctx, const int* xsrc, const int* ysrc, int nres, hipStream_t st,
                    const double* ystack, long long ystride)
{
    const int S = ctx->S, J = ctx->J, MTc = ceil_div(ctx->Tp, 16), KT = std::max(1, 12 / MTc);      // (two column tiles per wave)
    const int nks_c = round_up(ceil_div(S, 4), KT);
    const int npairs = nres * J;
    const MomLayout ml = moment_layout(ctx, npairs);
    const int groups_m = ml.groups;
    const size_t astride = (size_t)nks_c * MTc * 64, mstride = ml.stride;
    if (int e = ensure_compact_maps(ctx)) return e;
    if (int e = ensure(ctx, ctx->Afrag_c, (size_t)nres * astride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->mask_c, (size_t)nres * S)) return e;
    if (int e = ensure(ctx, ctx->rank_c, (size_t)nres * S * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->rowtab_c, ((size_t)nres * nks_c * 4 + nres) * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->Afrag_m, (size_t)groups_m * mstride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->momn_m, (size_t)round_up(npairs, 192) * 8)) return e;
    if (int e = ensure(ctx, ctx->scale, (size_t)round_up(npairs, 8) * ctx->Bpad * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->Afrag_c.p, 0, (size_t)nres * astride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->Afrag_m.p, 0, (size_t)groups_m * mstride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->mask_c.p, 0, (size_t)nres * S, st));
    int* row_cnt = ptr<int>(ctx->rowtab_c) + (size_t)nres * nks_c * 4;
    ctx->last_compact_n = nres; ctx->last_compact_ktot = nks_c * 4;
    {
        KTimer tm(ctx, KC_BUILD, st);
        hipLaunchKernelGGL(k_drawn_mask, dim3(ceil_div(S, 256), nres), dim3(256), 0, st, xsrc, S, ptr<uint8_t>(ctx->mask_c));
        LAUNCHCHK();
        hipLaunchKernelGGL(k_split_rank, dim3(nres), dim3(64), 0, st, ptr<uint8_t>(ctx->mask_c), S, nks_c * 4,
                           ptr<int>(ctx->rank_c), ptr<int>(ctx->rowtab_c), row_cnt);
        LAUNCHCHK();
        GroupLayout lay;
        lay.n = 1; lay.Tp = ctx->Tp; lay.J = J; lay.T = ctx->T; lay.MT = MTc; lay.w0 = MTc; lay.sq0 = MTc; lay.Tpp = ctx->Tpp;
        hipLaunchKernelGGL(k_build_A_behav, dim3(nres, J), dim3(256), (size_t)2 * ctx->T * 8, st,
                           ystack ? ystack : ptr<double>(ctx->Y), ystack ? ystride : 0LL, ctx->T, S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), xsrc, ysrc, lay, ctx->cov, 1,
                           ptr<double>(ctx->Afrag_c), astride, ptr<double>(ctx->momn_m), 0, 0,
                           ptr<double>(ctx->Afrag_m), mstride, ptr<int>(ctx->rank_c), ml.pairs);
        LAUNCHCHK();
    }
    SplitEpi se;
    memset(&se, 0, sizeof(se));
    se.scale = ptr<double>(ctx->scale);
    se.npairs = npairs;
    if (int e = launch_moment_blocks<4>(ctx, ml, se, st)) return e;
    se.npairs = J;
    se.accB = nres * ctx->Tpp;
    se.row_tab = ptr<int>(ctx->rowtab_c);
    se.row_cnt = row_cnt;
    const bool tail = ctx->Tp - (MTc - 1) * 16 <= 4 && MTc >= 2;
    switch (MTc) {
        case 1: return launch_xprod_cboot<1, 12>(ctx, nres, nks_c, se, st);
        case 2: return tail ? launch_xprod_cboot<2, 6, true>(ctx, nres, nks_c, se, st)
                            : launch_xprod_cboot<2, 6>(ctx, nres, nks_c, se, st);
        case 3: return tail ? launch_xprod_cboot<3, 4, true>(ctx, nres, nks_c, se, st)
                            : launch_xprod_cboot<3, 4>(ctx, nres, nks_c, se, st);
        case 4: return tail ? launch_xprod_cboot<4, 3, true>(ctx, nres, nks_c, se, st)
                            : launch_xprod_cboot<4, 3>(ctx, nres, nks_c, se, st);
        // 64 < T' <= 208: 5 .. 13 tiles, 3 or 2 waves per SIMD (the accumulators of two column tiles)
#define PLSX_CB(M, K) case M: return tail ? launch_xprod_cboot<M, K, true>(ctx, nres, nks_c, se, st) \
                                          : launch_xprod_cboot<M, K>(ctx, nres, nks_c, se, st);
        PLSX_CB(5, 2) PLSX_CB(6, 2) PLSX_CB(7, 1) PLSX_CB(8, 1) PLSX_CB(9, 1) PLSX_CB(10, 1) PLSX_CB(11, 1) PLSX_CB(12, 1) PLSX_CB(13, 1)
#undef PLSX_CB
        default: return fail(ctx, PLSX_ERR_STATE, "compact bootstrap blocks: T' > 208");
    }
}

// Build the A operands of `nres` resamples and run the cross-product kernel:
// afterwards R[r] (r < nres) holds gen_covcorr of resample r in columns
// [0, B) and its gen_distrib in columns [B, B+L) (once the original is set).
// ystack: per-resample behaviour matrices (S x T each, `ystride` doubles apart; ystride 0 =
// one matrix shared by all resamples of the call, e.g. the halves of one pre-permuted Y).
int run_xprod(plsx_ctx* ctx, const int* xsrc, const int* ysrc, int nres, hipStream_t st,
              bool prebuilt = false, const double* ystack = nullptr, long long ystride = -1, bool sparse_rows = false)
{
    // sparse_rows: the caller's resamples draw a good part of the rows of X more than once or not at all
    // (bootstraps): compact blocks when the shape allows
    if (ystride < 0) ystride = (long long)ctx->S * ctx->T;
    const int groups = ceil_div(nres, ctx->npg);
    if (int e = ensure_scratch(ctx, groups)) return e;
    if (ctx->timing) ctx->timed_units += nres;
    const int pgroups = phys_groups(ctx, groups);
    ctx->last_compact_n = 0;
    if (prebuilt) return launch_xprod(ctx, pgroups, st);       // A already scattered by the caller
    if (sparse_rows && xsrc && compact_boot_ok(ctx)) return run_xprod_cboot(ctx, xsrc, ysrc, nres, st, ystack, ystride);
    if (ctx->sepmom && ctx->method == PLSX_BEHAVIORAL && !ctx->mom_out_arg) {
        // tile passes of the launch in either layout; the separate-moments layout has to win by 2 %
        // (one more launch, the scale table): it does at the headline shape (1728 -> 1656 per 504
        // bootstraps), not for a handful of resamples or for small T' (c2: 32 vs 38 resamples a block)
        const long long cost_a = (long long)groups * ctx->MT;
        const long long cost_b = (long long)ceil_div(nres, ctx->npg_d) * ctx->MTd +
                                 (long long)ceil_div(nres * ctx->J, PLSX_MOM_PAIRS) * 24;
        const bool force_b = ctx->opt[OPT_SEPMOM_ALWAYS] != 0;     // tests: the layout at any launch size
        ctx->sepmom_used = force_b || cost_b * 102 < cost_a * 100;
        if (ctx->sepmom_used)
            return run_xprod_sepmom(ctx, xsrc, ysrc, nres, st, ystack, ystride);
    }
    HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)pgroups * ctx->group_stride * 8, st));
    GroupLayout lay;
    lay.n = ctx->npg; lay.Tp = ctx->Tp; lay.J = ctx->J; lay.T = ctx->T; lay.MT = ctx->MT;
    lay.w0 = ctx->w0; lay.sq0 = ctx->sq0; lay.Tpp = ctx->Tpp;
    if (ctx->gps > 0) {
        lay.gps = ctx->gps; lay.row_slice = ptr<int>(ctx->row_slice); lay.row_local = ptr<int>(ctx->row_local);
        lay.slice_cell0 = ptr<int>(ctx->slice_cell0);
    }
    if (ctx->method == PLSX_REGRESSION) {
        // A (dual weights) was scattered by k_simpls_dual; nothing to build here
    } else if (ctx->method == PLSX_BEHAVIORAL) {
        dim3 grid(nres, ctx->J), block(256);
        const size_t lds = (size_t)2 * ctx->T * 8;
        hipLaunchKernelGGL(k_build_A_behav, grid, block, lds, st, ystack ? ystack : ptr<double>(ctx->Y),
                           ystack ? ystride : 0LL, ctx->T, ctx->S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), xsrc, ysrc, lay,
                           ctx->cov, ctx->momrows, ptr<double>(ctx->Afrag), ctx->group_stride,
                           ptr<double>(ctx->mom_n), std::max(ctx->nmom_pad, 16));
    } else {
        dim3 grid(nres), block(256);
        hipLaunchKernelGGL(k_build_A_mc, grid, block, 0, st, ctx->S, ctx->J, ctx->n_cond, ctx->mc,
                           ptr<int>(ctx->cell_of_row), xsrc, lay, ptr<double>(ctx->Afrag),
                           ctx->group_stride);
    }
    LAUNCHCHK();
    return launch_xprod(ctx, pgroups, st);
}

// Resident blocks of `kernel` (256-thread blocks) on the whole chip.
int chip_slots(const void* kernel)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1)
        per_cu = 2;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return per_cu * cus;
}

// Number of parts (lo..hi) to cut each of `units` work items into so that
// units * parts fills whole rounds of `slots` resident blocks as exactly as
// possible (a grid that ends in a nearly empty last round wastes up to a round).
int pick_parts(long long units, int slots, int lo, int hi)
{
    int best = lo;
    double best_waste = 2.0;
    for (int p = lo; p <= hi; ++p) {
        const double rounds = (double)units * p / slots;
        const double waste = (rounds < 1.0) ? 0.0 : (std::ceil(rounds) - rounds) / std::ceil(rounds);
        if (waste < best_waste - 1e-9) { best_waste = waste; best = p; }
        if (waste < 0.03) break;
    }
    return best;
}

// C1 = A.B1^T (and C2 = A.B2^T), batched, contraction over K columns.
int run_nt(plsx_ctx* ctx, const double* A, long long strideA, int lda, int Ma,
           const double* B1, long long strideB1, int ldb1, int N1,
           const double* B2, long long strideB2, int ldb2, int N2, int K, int batch,
           double* C1, long long strideC1, int ldc1, double* C2, long long strideC2, int ldc2,
           hipStream_t st, bool sym = false, bool accumulate = false)
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
                hipStream_t st, const double* Rsrc = nullptr)
{
    // Rsrc: the (nres x T'pp x Bpad) blocks to multiply when they are not the cross-product scratch itself
    // (cross-validation's rescaled copies)
    const double* R = Rsrc ? Rsrc : ptr<double>(ctx->R);
    double* Gm = ptr<double>(ctx->Gm);
    const long long sG = (long long)ctx->Tp * ctx->Tp, sP = (long long)ctx->Tp * Erows;
    if ((ctx->Tp > 64 || Erows > 64) && !ctx->opt[OPT_GRAM_NT]) {
        // 64 x 64 output blocks on the register-streamed k_gram (4 x the rate of the generic
        // LDS-tiled k_nt_gemm): G upper block triangle, then P, into one partial buffer
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
        const bool reg_streamed = ctx->opt[OPT_GRAM_REG] != 0;     // the A side from global memory in every wave
        if (!reg_streamed) {
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
        } else {
            if (mode == 1 && nt_l == nt_t) {
                // square: G and P of the blocks tm <= tn share their A fragments in one pass,
                // P of the blocks below the diagonal follows
                hipLaunchKernelGGL(k_gram<1>, dim3(nchunk, nres, nz_g), dim3(256), 0, st, R, ctx->strideR, ctx->Bpad,
                                   ctx->Tp, E, ctx->Bpad, Erows, ctx->B, cols, part, nres, pitch, tiles, nt_t, 1);
                LAUNCHCHK();
                if (nt_t > 1) {
                    hipLaunchKernelGGL(k_gram<2>, dim3(nchunk, nres, nt_t * (nt_t - 1) / 2), dim3(256), 0, st, R,
                                       ctx->strideR, ctx->Bpad, ctx->Tp, E, ctx->Bpad, Erows, ctx->B, cols, part, nres,
                                       pitch, tiles, nt_t, 2);
                    LAUNCHCHK();
                }
            } else {
                if (nz_g) {
                    hipLaunchKernelGGL(k_gram<0>, dim3(nchunk, nres, nz_g), dim3(256), 0, st, R, ctx->strideR, ctx->Bpad,
                                       ctx->Tp, (const double*)nullptr, ctx->Bpad, 0, ctx->B, cols, part, nres, pitch,
                                       tiles, nt_t, 1);
                    LAUNCHCHK();
                }
                if (nz_p) {
                    hipLaunchKernelGGL(k_gram<2>, dim3(nchunk, nres, nz_p), dim3(256), 0, st, R, ctx->strideR, ctx->Bpad,
                                       ctx->Tp, E, ctx->Bpad, Erows, ctx->B, cols, part, nres, pitch, tiles, nt_l, 0);
                    LAUNCHCHK();
                }
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
    if (ctx->Tp > 64 || Erows > 64) {
        if (mode == 2)
            return run_nt(ctx, R, ctx->strideR, ctx->Bpad, ctx->Tp, E, 0, ctx->Bpad, Erows,
                          nullptr, 0, 0, 0, ctx->B, nres, Pout, sP, Erows, nullptr, 0, 0, st);
        return run_nt(ctx, R, ctx->strideR, ctx->Bpad, ctx->Tp, R, ctx->strideR, ctx->Bpad, ctx->Tp,
                      mode == 1 ? E : nullptr, 0, ctx->Bpad, Erows, ctx->B, nres,
                      Gm, sG, ctx->Tp, mode == 1 ? Pout : nullptr, sP, Erows, st);
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

// Rref: the cross-covariance matrices the Gram matrices a.G were formed from (slot r at r * strideR, pitch Bpad,
// B live columns), or nullptr on the dual-space routes that never form them.  With them a graded spectrum is
// refined on R itself (SmallArgs::phase, k_refine_gram); without, such resamples are only counted.
int run_small(plsx_ctx* ctx, SmallArgs a, int nres, hipStream_t st, const double* Rref = nullptr)
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
        int nblk = 0;
#define SMALL_QL_LAUNCH(RPT, CH) { HIPCHK(set_lds(k_small_ql<RPT, CH>, lds)); int per = 1; \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_small_ql<RPT, CH>, PLSX_SE_THREADS, lds); \
        nblk = std::min(nres, 256 * std::max(1, per)); \
        if (int e = ensure(ctx, ctx->gws, (size_t)nblk * ws)) return e; \
        a.gws = ptr<double>(ctx->gws); \
        hipLaunchKernelGGL((k_small_ql<RPT, CH>), dim3(nblk), dim3(PLSX_SE_THREADS), lds, st, a); }
        if (n <= 192) SMALL_QL_LAUNCH(1, 16)          // rows of the eigenvector matrix per rotating thread, prefetch depth
        else if (n <= 384) SMALL_QL_LAUNCH(2, 8)
        else if (n <= 576) SMALL_QL_LAUNCH(3, 8)
        else SMALL_QL_LAUNCH(7, 4)
#undef SMALL_QL_LAUNCH
        LAUNCHCHK();
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

// Waves per block of the rotation kernel: every block copies the M operand of every resample to LDS, so 8
// waves (128 features) per block halve that L2 -> LDS stream (as large as the HBM stream of R at 4 waves) for
// the compiled-in k-step counts at large B (c4: 23.3 -> 22.1 ms per 1008 bootstraps); the generic variants and
// small B (c2: 0.97 vs 1.03 ms) keep 4.
inline int urot_waves(const plsx_ctx* ctx, int nks_template, int B)
{
    const bool four = ctx->opt[OPT_UROT_NW4] != 0;
    return (nks_template > 0 && !four && B >= 65536) ? 8 : 4;      // (few feature tiles: more, smaller blocks fill the chip)
}

// One launch of the rotation kernel for the chunk of L tiles [lt0, lt0 + LT).
template <int LT, int NKS, bool TAIL = false, int NST = 2>
int launch_urot(plsx_ctx* ctx, int nres, int lt0, int nsplit, int rps, double* usum, double* usq, double* out,
                double* ps, double* pq, hipStream_t st)
{
    const int nw = urot_waves(ctx, NKS, ctx->B);
    const int nblk = ceil_div(ceil_div(ctx->B, 16), nw);
    // NST LDS stages of the M operand (whole 1 KB DMA pieces); none when M stays in L2
    const size_t lds = (size_t)NST * ceil_div((NKS < 0 ? PLSX_UROT_KC : ctx->nks_t) * LT, 2) * 1024;
    HIPCHK(set_lds((k_urot<LT, NKS, TAIL, NST>), lds));
    const double* M = ptr<double>(ctx->Mfrag) + mfrag_chunk_base(lt0 / PLSX_LT_CHUNK, ctx->nks_t);
    hipLaunchKernelGGL((k_urot<LT, NKS, TAIL, NST>), dim3(nblk, nsplit), dim3(64 * nw), lds, st, ptr<double>(ctx->R),
                       ctx->strideR, ctx->Bpad, ctx->nks_t, M, (size_t)ctx->nks_t * ctx->LT * 64, nres, ctx->B,
                       ctx->L, lt0 * 16, usum, usq, out, rps, ps, pq);
    LAUNCHCHK();
    return 0;
}

template <int NKS>
int launch_urot_lt(plsx_ctx* ctx, int ltc, int nres, int lt0, int nsplit, int rps, double* usum, double* usq,
                   double* out, double* ps, double* pq, hipStream_t st)
{
    switch (ltc) {
        case 1: return launch_urot<1, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        case 2: return launch_urot<2, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        case 3: return launch_urot<3, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        case 4: return launch_urot<4, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        case 5: return launch_urot<5, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        default: return launch_urot<6, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
    }
}

int run_urot(plsx_ctx* ctx, int nres, double* usum, double* usq, double* out, hipStream_t st)
{
    const int nks = ctx->nks_t, LT = ctx->LT;
    KTimer tm(ctx, KC_UROT, st);
    const bool square = !ctx->opt[OPT_UROT_GENERIC] && LT <= PLSX_LT_CHUNK && LT == ceil_div(nks, 4) && nks <= 16;
    const int nblk = ceil_div(ceil_div(ctx->B, 16), urot_waves(ctx, square ? 1 : 0, ctx->B));
    int nsplit = 1;
    if (!out && nres >= 64) {
        const int slots = std::max(1, chip_slots(reinterpret_cast<const void*>(k_urot<4, 0>)) * 4 / urot_waves(ctx, square ? 1 : 0, ctx->B));
        // many feature blocks: cut the resamples so that the grid ends in a full round;
        // few (small B): cut them so that the grid fills the chip at all -- every block
        // walks its resamples one after the other
        nsplit = nblk >= slots ? pick_parts(nblk, slots, 1, 8) : std::min(32, ceil_div(2 * slots, nblk));
        nsplit = std::max(1, std::min(nsplit, nres / 32));
    }
    const int rps = ceil_div(nres, std::max(nsplit, 1));
    nsplit = ceil_div(nres, rps);
    double *ps = nullptr, *pq = nullptr;
    if (nsplit > 1) {
        const size_t bytes = (size_t)nsplit * ctx->B * ctx->L * 8;
        if (int e = ensure(ctx, ctx->psum, bytes)) return e;
        if (int e = ensure(ctx, ctx->psq, bytes)) return e;
        ps = ptr<double>(ctx->psum);
        pq = ptr<double>(ctx->psq);
    }
    const bool generic = ctx->opt[OPT_UROT_GENERIC] != 0;   // A/B and race check
    int rc = -1;
    // square case (L tiles follow from T'): k-step count compiled in, fragments of the
    // next resample prefetched
    // the last tile of L on the 4x4x4 shape when it holds at most 4 live columns (see k_urot)
    const bool tail4 = ctx->L - 16 * (LT - 1) <= 4 && !ctx->opt[OPT_UROT_NO_TAIL4];
    if (!generic && ctx->opt[OPT_UROT_M3] && nks == 13 && LT == 4 && tail4)
        rc = launch_urot<4, 13, true, 3>(ctx, nres, 0, nsplit, rps, usum, usq, out, ps, pq, st);
    else if (!generic && LT <= PLSX_LT_CHUNK && LT == ceil_div(nks, 4)) {
        switch (nks) {
#define UCASE(N) case N: rc = tail4 ? launch_urot<(N + 3) / 4, N, true>(ctx, nres, 0, nsplit, rps, usum, usq, out, ps, pq, st) \
                                    : launch_urot<(N + 3) / 4, N>(ctx, nres, 0, nsplit, rps, usum, usq, out, ps, pq, st); break;
        UCASE(1) UCASE(2) UCASE(3) UCASE(4) UCASE(5) UCASE(6) UCASE(7) UCASE(8)
        UCASE(9) UCASE(10) UCASE(11) UCASE(12) UCASE(13) UCASE(14) UCASE(15) UCASE(16)
#undef UCASE
        default: break;
        }
    }
    if (rc < 0) {
        // generic: one launch per chunk of PLSX_LT_CHUNK tiles; M through LDS while two
        // stages of a chunk fit (150 KB), from L2 otherwise
        rc = 0;
        for (int lt0 = 0; lt0 < LT && rc == 0; lt0 += PLSX_LT_CHUNK) {
            const int ltc = std::min(PLSX_LT_CHUNK, LT - lt0);
            const bool in_lds = (size_t)2 * ceil_div(nks * ltc, 2) * 1024 <= 150 * 1024;
            rc = in_lds ? launch_urot_lt<0>(ctx, ltc, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st)
                        : launch_urot_lt<-1>(ctx, ltc, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        }
    }
    if (rc) return rc;
    if (nsplit > 1) {
        const long long count = (long long)ctx->B * ctx->L;
        hipLaunchKernelGGL(k_add_splits, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ps, pq, nsplit,
                           count, usum, usq);
        LAUNCHCHK();
    }
    return 0;
}

// Split-half feature-axis sums: same chunking of L.
template <int LT, int NKS, bool TAIL = false, class... Args>
int launch_ucorr_t(plsx_ctx* ctx, dim3 grid, dim3 block, hipStream_t st, Args... args)
{
    const size_t lds = NKS < 0 ? 0 : (size_t)ctx->nks_t * LT * 64 * 8;
    HIPCHK(set_lds(k_ucorr_partial<LT, NKS, TAIL>, lds));
    hipLaunchKernelGGL((k_ucorr_partial<LT, NKS, TAIL>), grid, block, lds, st, args...);
    return 0;
}

template <int NKS, class... Args>
int launch_ucorr_lt(plsx_ctx* ctx, int ltc, dim3 grid, dim3 block, hipStream_t st, Args... args)
{
    switch (ltc) {
        case 1: return launch_ucorr_t<1, NKS>(ctx, grid, block, st, args...);
        case 2: return launch_ucorr_t<2, NKS>(ctx, grid, block, st, args...);
        case 3: return launch_ucorr_t<3, NKS>(ctx, grid, block, st, args...);
        case 4: return launch_ucorr_t<4, NKS>(ctx, grid, block, st, args...);
        case 5: return launch_ucorr_t<5, NKS>(ctx, grid, block, st, args...);
        default: return launch_ucorr_t<6, NKS>(ctx, grid, block, st, args...);
    }
}

// M (fragment order, chunked) for all of L; partial sums [nchunk][npairs][5][lpad]
int launch_ucorr(plsx_ctx* ctx, dim3 grid, dim3 block, hipStream_t st, const double* M, int tpc,
                 double* part, int npairs)
{
    const int nks = ctx->nks_t, LT = ctx->LT, lpad = LT * 16;
    const double* R = ptr<double>(ctx->R);
    KTimer tm(ctx, KC_UCORR, st);
    const bool tail4 = ctx->L - 16 * (LT - 1) <= 4 && !ctx->opt[OPT_UROT_NO_TAIL4];
    if (LT <= PLSX_LT_CHUNK && LT == ceil_div(nks, 4)) {
        switch (nks) {
#define UCASE(N) case N: return tail4 ? launch_ucorr_t<(N + 3) / 4, N, true>(ctx, grid, block, st, R, ctx->strideR, ctx->Bpad, \
                    nks, M, ctx->B, tpc, part, npairs, 0, lpad) \
                                     : launch_ucorr_t<(N + 3) / 4, N>(ctx, grid, block, st, R, ctx->strideR, ctx->Bpad, \
                    nks, M, ctx->B, tpc, part, npairs, 0, lpad);
        UCASE(1) UCASE(2) UCASE(3) UCASE(4) UCASE(5) UCASE(6) UCASE(7) UCASE(8)
        UCASE(9) UCASE(10) UCASE(11) UCASE(12) UCASE(13) UCASE(14) UCASE(15) UCASE(16)
#undef UCASE
        default: break;
        }
    }
    for (int lt0 = 0; lt0 < LT; lt0 += PLSX_LT_CHUNK) {
        const int ltc = std::min(PLSX_LT_CHUNK, LT - lt0);
        const double* Mc = M + mfrag_chunk_base(lt0 / PLSX_LT_CHUNK, nks);
        const bool in_lds = (size_t)nks * ltc * 512 <= 128 * 1024;       // + 15 KB of static reduction space
        const int rc = in_lds ? launch_ucorr_lt<0>(ctx, ltc, grid, block, st, R, ctx->strideR, ctx->Bpad, nks, Mc,
                                                   ctx->B, tpc, part, npairs, lt0 * 16, lpad)
                              : launch_ucorr_lt<-1>(ctx, ltc, grid, block, st, R, ctx->strideR, ctx->Bpad, nks, Mc,
                                                    ctx->B, tpc, part, npairs, lt0 * 16, lpad);
        if (rc) return rc;
    }
    return 0;
}

SmallArgs small_args(plsx_ctx* ctx, int mode)
{
    SmallArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = mode; a.n = ctx->Tp; a.L = ctx->L; a.rotate = 1;
    a.G = ptr<double>(ctx->Gm); a.P = ptr<double>(ctx->Pm);
    a.V0 = ptr<double>(ctx->V0); a.d0 = ptr<double>(ctx->d0);
    a.Mfrag = ptr<double>(ctx->Mfrag); a.nks_t = ctx->nks_t; a.LT = ctx->LT;
    a.status = ptr<int>(ctx->status);
    return a;
}

bool plsc_single_pass(const plsx_ctx* ctx);
// Dual-space routes (S x S kernel: permutations, single-pass bootstraps of the unscaled modes) never form R and
// so cannot refine a graded spectrum (run_small); a data set whose ORIGINAL spectrum is graded takes the feature pass.
inline int use_dual(const plsx_ctx* ctx) { return (ctx->dual && !ctx->graded) ? 1 : 0; }

bool plsc_single_pass(const plsx_ctx* ctx)
{
    return ctx->method != PLSX_REGRESSION && !ctx->scaled && use_dual(ctx) && ctx->gps == 0 && ctx->L == ctx->Tp &&
           ctx->Tp <= PLSX_JACOBI_TP && 2 * (size_t)ctx->L * PLSX_ACC_PITCH * 8 <= 72 * 1024 && !ctx->opt[OPT_TWO_PASS_BOOT];
}

// d (L values on the device, descending): set ctx->graded when a live singular value lies below PLSX_REFINE_TAU d_max.
int note_spectrum(plsx_ctx* ctx, const double* d_sv, hipStream_t st)
{
    std::vector<double> d(ctx->L);
    HIPCHK(hipMemcpyAsync(d.data(), d_sv, (size_t)ctx->L * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double dmax = 0.0;
    for (double v : d) if (v > dmax) dmax = v;
    int graded = 0;
    for (double v : d) if (v > PLSX_RANK_RTOL * dmax && v < PLSX_REFINE_TAU * dmax) graded = 1;
    if (graded != ctx->graded) ctx->has_Kd = ctx->has_Kd && !graded;
    ctx->graded = graded && !ctx->opt[OPT_NO_REFINE];
    return 0;
}

#define NEED_DATA()                                                             \
    if (!ctx) return PLSX_ERR_ARG;                                              \
    if (!ctx->has_data) return fail(ctx, PLSX_ERR_STATE, "plsx_set_data has not been called")
#define NEED_ORIG()                                                             \
    NEED_DATA();                                                                \
    if (!ctx->has_orig) return fail(ctx, PLSX_ERR_STATE, "plsx_set_original has not been called")

}  // namespace

extern "C" {

int plsx_version(void) { return 1000; }
int plsx_max_tprime(void) { return PLSX_MAX_TP; }

int plsx_ctx_create(int device, plsx_ctx** out)
try {
    if (!out) return PLSX_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return PLSX_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return PLSX_ERR_HIP;
    plsx_ctx* c = new (std::nothrow) plsx_ctx();
    if (!c) return PLSX_ERR_HIP;
    c->device = device;
    if (hipMalloc(&c->status.p, 4 * sizeof(int)) != hipSuccess || hipMemset(c->status.p, 0, 4 * sizeof(int)) != hipSuccess) {
        delete c;
        return PLSX_ERR_HIP;
    }
    c->status.bytes = 4 * sizeof(int);
    *out = c;
    return PLSX_OK;
} PLSX_CATCH(nullptr)

int plsx_ctx_destroy(plsx_ctx* ctx)
try {
    if (!ctx) return PLSX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (Buf* b : {&ctx->Xc, &ctx->xmean, &ctx->Y, &ctx->cell_of_row, &ctx->cell_start, &ctx->cell_len,
                   &ctx->out_row, &ctx->mom_idx, &ctx->mom_n, &ctx->Afrag, &ctx->R, &ctx->Gm, &ctx->Pm,
                   &ctx->part, &ctx->Mfrag, &ctx->U0T, &ctx->V0, &ctx->d0, &ctx->tmpW,
                   &ctx->Rfull, &ctx->Vp, &ctx->dp, &ctx->Mvd, &ctx->Cm, &ctx->srcx, &ctx->srcy, &ctx->part2,
                   &ctx->Kmat, &ctx->swork, &ctx->spct, &ctx->sc,
                   &ctx->momout, &ctx->R2, &ctx->cvc, &ctx->Qm, &ctx->Vs, &ctx->ds, &ctx->ybar, &ctx->pred,
                   &ctx->Xn, &ctx->out_row_f, &ctx->mom_idx_f, &ctx->Kd, &ctx->Ad, &ctx->Wd, &ctx->gws, &ctx->cellS, &ctx->rowc, &ctx->out_row_s, &ctx->okx, &ctx->oky, &ctx->psum, &ctx->psq, &ctx->row_slice, &ctx->row_local, &ctx->slice_cell0, &ctx->cell_momrow, &ctx->status, &ctx->ScT, &ctx->out_row_w, &ctx->Qs, &ctx->out_row_d, &ctx->mom_idx_d, &ctx->Afrag_m, &ctx->momn_m, &ctx->scale,
                   &ctx->Afrag_c, &ctx->rank_c, &ctx->rowtab_c, &ctx->m1_c, &ctx->m2_c, &ctx->out_row_c, &ctx->mom_idx_c, &ctx->mask_c,
                   &ctx->refV, &ctx->refLam, &ctx->refK0, &ctx->refPart, &ctx->refPartP, &ctx->refH, &ctx->flipws, &ctx->pflags,
                   &ctx->Cq, &ctx->Vsumq, &ctx->Vdq, &ctx->Vtq, &ctx->Afrag_q, &ctx->qpart})
        release(*b);
    for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); }
    delete ctx;
    return PLSX_OK;
} PLSX_CATCH(ctx)

const char* plsx_last_error(const plsx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int plsx_sync(plsx_ctx* ctx)
try {
    if (!ctx) return PLSX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    // numerical status of everything that ran since the last call: an eigen-solve that gave up is an
    // error of the results already written, reported here instead of flowing on silently
    int stw[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(stw, ctx->status.p, 4 * sizeof(int), hipMemcpyDeviceToHost));
    if (stw[1] || stw[2]) {
        ctx->n_refined += stw[1];
        ctx->n_unrefined += stw[2];
        HIPCHK(hipMemset(static_cast<int*>(ctx->status.p) + 1, 0, 2 * sizeof(int)));
    }
    const int st = stw[0];
    if (st) {
        HIPCHK(hipMemset(ctx->status.p, 0, sizeof(int)));
        return fail(ctx, PLSX_ERR_NUMERIC, "small solver: implicit QL did not converge within 60 iterations for at least "
                                           "one resample (non-finite or pathological Gram matrix); results of the batch are invalid");
    }
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_num_lv(const plsx_ctx* ctx) { return (ctx && ctx->has_data) ? ctx->L : PLSX_ERR_STATE; }
int plsx_tprime(const plsx_ctx* ctx) { return (ctx && ctx->has_data) ? ctx->Tp : PLSX_ERR_STATE; }

int plsx_set_data(plsx_ctx* ctx, int method, const double* d_X, const double* d_Y,
                  const int32_t* d_cell_of_row, int S, int B, int T, int n_groups, int n_cond,
                  int mean_centering, unsigned flags, void* stream)
try {
    if (!ctx) return PLSX_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    if (!d_X || !d_cell_of_row || S < 2 || B < 1 || n_groups < 1 || n_cond < 1)
        return fail(ctx, PLSX_ERR_ARG, "plsx_set_data: bad shape or null pointer");
    if (method != PLSX_BEHAVIORAL && method != PLSX_MEANCENTERED && method != PLSX_REGRESSION)
        return fail(ctx, PLSX_ERR_ARG, "plsx_set_data: unknown method");
    if (method != PLSX_MEANCENTERED && (!d_Y || T < 1))
        return fail(ctx, PLSX_ERR_ARG, "plsx_set_data: this method needs Y");
    // PLSX_REGRESSION: n_cond carries n_components (rows of x_weights^T per resample)
    const int ncomp = (method == PLSX_REGRESSION) ? n_cond : 0;
    if (method == PLSX_REGRESSION) {
        if (ncomp < 1 || ncomp > std::min(S - 1, B))
            return fail(ctx, PLSX_ERR_ARG, "plsx_set_data: n_components out of range");
        n_groups = 1; n_cond = 1;
    }
    if (mean_centering < 0 || mean_centering > 2)
        return fail(ctx, PLSX_ERR_ARG, "plsx_set_data: mean_centering must be 0, 1 or 2");
    const int J = n_groups * n_cond;
    const int Tp = (method == PLSX_BEHAVIORAL) ? J * T : (method == PLSX_REGRESSION ? ncomp : J);
    if (method == PLSX_REGRESSION && sd_step_lds(S, T, ncomp) * 8 > 158 * 1024)
        return fail(ctx, PLSX_ERR_UNSUPPORTED,
                    "SIMPLS: S / T too large for the on-chip component step (8 (T^2 + S) bytes must fit 158 KB)");
    if ((long long)B + Tp > 2000000LL)
        return fail(ctx, PLSX_ERR_UNSUPPORTED, "more than 2,000,000 feature columns (32-bit buffer offsets)");
    if (Tp > PLSX_MAX_TP || J > PLSX_MAX_CELLS || (method != PLSX_BEHAVIORAL && Tp > PLSX_BLOCK_TP)) {
        char msg[200];
        snprintf(msg, sizeof msg, "stacked dimension T' = %d (cells J = %d) exceeds the limit (T' <= %d for "
                 "behavioral PLS, %d otherwise; J <= %d)", Tp, J, PLSX_MAX_TP, PLSX_BLOCK_TP, PLSX_MAX_CELLS);
        return fail(ctx, PLSX_ERR_UNSUPPORTED, msg);
    }
    ctx->has_data = ctx->has_orig = false;
    ctx->has_Kd = 0;
    ctx->npg_w = 0;                                    // the row -> LV map of the accumulating epilogue follows L
    ctx->has_compact_maps = 0;
    // a re-bound context keeps its scratch: the padding rows (t >= T') of every R slot must
    // read as zero under the new layout too
    {
        // ... unless the layout of a slot is the one the buffer was last zeroed under (the cached engine of the
        // front-ends re-binding data of the same shape: a 26 GB fill is 8 ms per call)
        const int Tp_n = (method == PLSX_BEHAVIORAL) ? J * T : (method == PLSX_REGRESSION ? ncomp : J);
        const int L_n = std::min(Tp_n, B);
        const long long geom[3] = {Tp_n, round_up(Tp_n, 4), round_up(B + L_n, 128)};
        const bool same = ctx->R.p && ctx->R_geom[0] == geom[0] && ctx->R_geom[1] == geom[1] && ctx->R_geom[2] == geom[2] &&
                          ctx->R_zeroed_bytes == ctx->R.bytes;
        if (ctx->R.p && !same) HIPCHK(hipMemsetAsync(ctx->R.p, 0, ctx->R.bytes, st));
        ctx->R_geom[0] = geom[0]; ctx->R_geom[1] = geom[1]; ctx->R_geom[2] = geom[2];
        ctx->R_zeroed_bytes = ctx->R.bytes;
    }
    ctx->has_okx = ctx->has_oky = false;
    ctx->Galloc = 0;
    ctx->method = method; ctx->S = S; ctx->B = B; ctx->T = (method == PLSX_MEANCENTERED) ? 0 : T;
    ctx->ncomp = ncomp;
    ctx->J = J; ctx->n_groups = n_groups; ctx->n_cond = n_cond; ctx->mc = mean_centering;
    ctx->cov = (flags & PLSX_FLAG_COVARIANCE) ? 1 : 0;
    ctx->Tp = Tp; ctx->Tpp = round_up(Tp, 4); ctx->L = std::min(Tp, B);
    ctx->Kpad = round_up(S, 8); ctx->nks = ctx->Kpad / 4;
    ctx->Bx = B + ctx->L; ctx->Bpad = round_up(ctx->Bx, 128);
    ctx->nks_t = ctx->Tpp / 4; ctx->LT = ceil_div(ctx->L, 16);
    ctx->strideR = (long long)ctx->Tpp * ctx->Bpad;
    ctx->MT = 24;

    // cell layout (host copy): cells must be contiguous row ranges (pyls/utils.py:178-197)
    std::vector<int> cells(S);
    HIPCHK(hipMemcpy(cells.data(), d_cell_of_row, S * sizeof(int), hipMemcpyDeviceToHost));
    ctx->h_cell_start.assign(J, 0);
    ctx->h_cell_len.assign(J, 0);
    for (int i = 0; i < S; ++i) {
        int c = cells[i];
        if (c < 0 || c >= J || (i > 0 && c < cells[i - 1]))
            return fail(ctx, PLSX_ERR_ARG, "plsx_set_data: cell_of_row must be non-decreasing in [0, J)");
        if (ctx->h_cell_len[c] == 0) ctx->h_cell_start[c] = i;
        ctx->h_cell_len[c]++;
    }
    for (int c = 0; c < J; ++c)
        if (ctx->h_cell_len[c] < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_set_data: empty cell");

    if (int e = ensure(ctx, ctx->cell_of_row, S * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->cell_start, J * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->cell_len, J * sizeof(int))) return e;
    HIPCHK(hipMemcpy(ctx->cell_of_row.p, cells.data(), S * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->cell_start.p, ctx->h_cell_start.data(), J * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->cell_len.p, ctx->h_cell_len.data(), J * sizeof(int), hipMemcpyHostToDevice));

    const size_t xbytes = (size_t)ctx->Kpad * ctx->Bpad * 8;
    if (int e = ensure(ctx, ctx->Xc, xbytes)) return e;
    if (int e = ensure(ctx, ctx->xmean, (size_t)ctx->Bpad * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->Xc.p, 0, xbytes, st));
    HIPCHK(hipMemsetAsync(ctx->xmean.p, 0, (size_t)ctx->Bpad * 8, st));
    hipLaunchKernelGGL(k_colmean, dim3(ceil_div(B, 256)), dim3(256), 0, st, d_X, S, B, ptr<double>(ctx->xmean));
    LAUNCHCHK();
    hipLaunchKernelGGL(k_center_pad, dim3(ceil_div(B, 256), S), dim3(256), 0, st, d_X,
                       ptr<double>(ctx->xmean), S, B, ptr<double>(ctx->Xc), ctx->Bpad);
    LAUNCHCHK();
    if (method != PLSX_MEANCENTERED) {
        if (int e = ensure(ctx, ctx->Y, (size_t)S * T * 8)) return e;
        HIPCHK(hipMemcpyAsync(ctx->Y.p, d_Y, (size_t)S * T * 8, hipMemcpyDeviceToDevice, st));
    }
    if (plan_groups(ctx) != 0)
        return fail(ctx, PLSX_ERR_UNSUPPORTED,
                    "cannot lay out the rows of a resample (with their per-cell moment rows) over cross-product blocks");
    if (int e = upload_rowmaps(ctx)) return e;
    if (int e = plan_sepmom(ctx)) return e;
    ctx->fix = 0; ctx->has_Xn = 0; ctx->npgf = 0; ctx->group_stride_f = 0; ctx->has_cellS = 0;
    {
        if (method == PLSX_BEHAVIORAL && !ctx->cov && !ctx->opt[OPT_NO_FIXED_X]) {
            // fixed-X fast path for permutations
            ctx->npgf = ctx->gps > 0 ? 0 : (ctx->MTf * 16) / ctx->Tp;
            {
                ctx->group_stride_f = (size_t)ctx->nks * ctx->MTf * 64;
                const int rows = ctx->MTf * 16;
                std::vector<int> orow(rows, -1), none(rows, -1);
                for (int rr = 0; rr < ctx->npgf; ++rr)
                    for (int t = 0; t < ctx->Tp; ++t) orow[rr * ctx->Tp + t] = rr * ctx->Tpp + t;
                if (ctx->npgf < 1) ctx->group_stride_f = 0;      // no fixed-X kernel: Xn only feeds the dual path
                if (int e = ensure(ctx, ctx->out_row_f, rows * sizeof(int))) return e;
                if (int e = ensure(ctx, ctx->mom_idx_f, rows * sizeof(int))) return e;
                HIPCHK(hipMemcpy(ctx->out_row_f.p, orow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(ctx->mom_idx_f.p, none.data(), rows * sizeof(int), hipMemcpyHostToDevice));
                if (int e = ensure(ctx, ctx->Xn, xbytes)) return e;
                HIPCHK(hipMemsetAsync(ctx->Xn.p, 0, xbytes, st));
                hipLaunchKernelGGL(k_cell_scale, dim3(ceil_div(B, 256)), dim3(256), 0, st, ptr<double>(ctx->Xc),
                                   ctx->Bpad, B, J, ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len),
                                   ptr<double>(ctx->Xn));
                LAUNCHCHK();
                ctx->has_Xn = 1;
                ctx->fix = ctx->npgf >= 1 ? 1 : 0;
            }
        }
    }
    {
        // dual permutation path: needs a resample-independent feature matrix
        ctx->dual_ok = (method == PLSX_MEANCENTERED || (method == PLSX_BEHAVIORAL && (ctx->has_Xn || ctx->cov))) ? 1 : 0;
        ctx->dual = (ctx->dual_ok && !ctx->opt[OPT_NO_DUAL_PERM]) ? 1 : 0;
        ctx->graded = 0;
    }
    if (int e = ensure(ctx, ctx->U0T, (size_t)ctx->L * ctx->Bpad * 8, true)) return e;
    if (int e = ensure(ctx, ctx->V0, (size_t)ctx->Tp * ctx->L * 8)) return e;
    if (int e = ensure(ctx, ctx->d0, (size_t)ctx->L * 8)) return e;
    if (method == PLSX_REGRESSION) {
        // K = Xc Xc^T (S x S): the only B-sized work the dual-space SIMPLS solver needs
        if (int e = ensure(ctx, ctx->Kmat, (size_t)S * S * 8)) return e;
        if (int e = run_nt(ctx, ptr<double>(ctx->Xc), 0, ctx->Bpad, S, ptr<double>(ctx->Xc), 0, ctx->Bpad, S,
                           nullptr, 0, 0, 0, B, 1, ptr<double>(ctx->Kmat), 0, S, nullptr, 0, 0, st, true))
            return e;
    }
    HIPCHK(hipStreamSynchronize(st));
    ctx->has_data = true;
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_colmean(plsx_ctx* ctx, double* d_mean, void* stream)
try {
    NEED_DATA();
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_mean, ctx->xmean.p, (size_t)ctx->B * 8, hipMemcpyDeviceToDevice,
                          static_cast<hipStream_t>(stream)));
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_crosscov_batch(plsx_ctx* ctx, const int32_t* d_xsrc, const int32_t* d_ysrc, int n,
                        double* d_R, void* stream)
try {
    NEED_DATA();
    if (n < 1 || !d_R) return fail(ctx, PLSX_ERR_ARG, "plsx_crosscov_batch: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    const int nb = launch_groups(ctx, n, ctx->npg) * ctx->npg;
    for (int off = 0; off < n; off += nb) {
        const int m = std::min(nb, n - off);
        const int* xs = d_xsrc ? d_xsrc + (size_t)off * ctx->S : nullptr;
        const int* ys = d_ysrc ? d_ysrc + (size_t)off * ctx->S : nullptr;
        if (int e = run_xprod(ctx, xs, ys, m, st)) return e;
        dim3 g(ceil_div(ctx->Tp * ctx->B, 256), m);
        hipLaunchKernelGGL(k_gather_cols, g, dim3(256), 0, st, ptr<double>(ctx->R), ctx->strideR,
                           ctx->Bpad, 0, ctx->Tp, ctx->B, d_R + (size_t)off * ctx->Tp * ctx->B);
        LAUNCHCHK();
    }
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_decompose(plsx_ctx* ctx, double* d_xw, double* d_sv, double* d_yw, void* stream)
try {
    NEED_DATA();
    if (!d_xw || !d_sv || !d_yw) return fail(ctx, PLSX_ERR_ARG, "plsx_decompose: null output");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    if (int e = run_xprod(ctx, nullptr, nullptr, 1, st)) return e;
    if (int e = run_gram(ctx, 1, false, st)) return e;
    SmallArgs a = small_args(ctx, SMALL_DECOMP);
    a.out_V = d_yw; a.out_d = d_sv;
    const bool fix = ctx->Tp <= PLSX_JACOBI_TP && ctx->Tp > 1 && !ctx->opt[OPT_NO_REFINE];
    if (fix) {
        if (int e = ensure(ctx, ctx->refH, (size_t)ctx->L * ctx->L * 8)) return e;
        a.out_H = ptr<double>(ctx->refH);
    }
    if (int e = run_small(ctx, a, 1, st, ptr<double>(ctx->R))) return e;
    if (int e = run_urot(ctx, 1, nullptr, nullptr, d_xw, st)) return e;
    if (fix) {
        // graded spectrum: the small x_weights columns lose their components along the large ones
        hipLaunchKernelGGL(k_fix_small_cols, dim3(ceil_div(ctx->B, 256)), dim3(256), 0, st, d_xw, ctx->B, ctx->L,
                           ptr<double>(ctx->refH), ptr<int>(ctx->refK0));
        LAUNCHCHK();
    }
    return note_spectrum(ctx, d_sv, st);
} PLSX_CATCH(ctx)

int plsx_project(plsx_ctx* ctx, const double* d_W, int L, double* d_out, void* stream)
try {
    NEED_DATA();
    if (!d_W || !d_out || L < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_project: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    if (int e = ensure(ctx, ctx->tmpW, (size_t)L * ctx->Bpad * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->tmpW.p, 0, (size_t)L * ctx->Bpad * 8, st));
    hipLaunchKernelGGL(k_transpose, dim3(ceil_div(L, 32), ceil_div(ctx->B, 32)), dim3(32, 8), 0, st,
                       d_W, ctx->B, L, L, ptr<double>(ctx->tmpW), ctx->Bpad);
    LAUNCHCHK();
    return run_nt(ctx, ptr<double>(ctx->Xc), 0, ctx->Bpad, ctx->S, ptr<double>(ctx->tmpW), 0, ctx->Bpad, L,
                  nullptr, 0, 0, 0, ctx->B, 1, d_out, 0, L, nullptr, 0, 0, st);
} PLSX_CATCH(ctx)

int plsx_set_original(plsx_ctx* ctx, const double* d_xw, const double* d_sv, const double* d_yw,
                      void* stream)
try {
    NEED_DATA();
    if (!d_xw || !d_sv || !d_yw) return fail(ctx, PLSX_ERR_ARG, "plsx_set_original: null input");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(ctx->V0.p, d_yw, (size_t)ctx->Tp * ctx->L * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->d0.p, d_sv, (size_t)ctx->L * 8, hipMemcpyDeviceToDevice, st));
    if (int e = note_spectrum(ctx, d_sv, st)) return e;
    hipLaunchKernelGGL(k_transpose, dim3(ceil_div(ctx->L, 32), ceil_div(ctx->B, 32)), dim3(32, 8), 0, st,
                       d_xw, ctx->B, ctx->L, ctx->L, ptr<double>(ctx->U0T), ctx->Bpad);
    LAUNCHCHK();
    // centred scores (X - mean) @ normalize(U0) into the extra columns [B, B+L)
    // of the feature matrix: the cross-product kernel then yields gen_distrib
    // (behavioral.py:78-80, meancentered.py:97-102) as L extra columns of R.
    // U0 columns are unit norm (or zero for null LVs), so normalize() is the identity.
    if (int e = run_nt(ctx, ptr<double>(ctx->Xc), 0, ctx->Bpad, ctx->S, ptr<double>(ctx->U0T), 0, ctx->Bpad,
                       ctx->L, nullptr, 0, 0, 0, ctx->B, 1, ptr<double>(ctx->Xc) + ctx->B, 0, ctx->Bpad,
                       nullptr, 0, 0, st))
        return e;
    // scores^T (L x S, pitch round_up(S, 8)): the B operand of P_r = A_r . scores in the single-pass
    // bootstrap of the unscaled modes (boot_single_pass)
    {
        const int Sd = round_up(ctx->S, 8);
        if (int e = ensure(ctx, ctx->ScT, (size_t)ctx->L * Sd * 8, true)) return e;
        HIPCHK(hipMemsetAsync(ctx->ScT.p, 0, (size_t)ctx->L * Sd * 8, st));
        hipLaunchKernelGGL(k_transpose, dim3(ceil_div(ctx->L, 32), ceil_div(ctx->S, 32)), dim3(32, 8), 0, st,
                           ptr<double>(ctx->Xc) + ctx->B, ctx->S, ctx->L, ctx->Bpad, ptr<double>(ctx->ScT), Sd);
        LAUNCHCHK();
    }
    ctx->has_orig = true; ctx->quad_active = 0;
    return PLSX_OK;
} PLSX_CATCH(ctx)

namespace {
int perm_batch_impl(plsx_ctx* ctx, const int32_t* d_perm_idx, const double* d_ystack, int n, int rotate,
                    double* d_out_sv, void* stream);
}

int plsx_perm_batch(plsx_ctx* ctx, const int32_t* d_perm_idx, int n, int rotate, double* d_out_sv,
                    void* stream)
try {
    NEED_ORIG();
    if (!d_perm_idx || !d_out_sv || n < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_perm_batch: bad arguments");
    return perm_batch_impl(ctx, d_perm_idx, nullptr, n, rotate, d_out_sv, stream);
} PLSX_CATCH(ctx)

int plsx_perm_batch_y(plsx_ctx* ctx, const double* d_ystack, int n, int rotate, double* d_out_sv,
                      void* stream)
try {
    NEED_ORIG();
    if (ctx->method != PLSX_BEHAVIORAL)
        return fail(ctx, PLSX_ERR_ARG, "plsx_perm_batch_y: pre-permuted Y stacks need behavioral PLS");
    if (!d_ystack || !d_out_sv || n < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_perm_batch_y: bad arguments");
    return perm_batch_impl(ctx, nullptr, d_ystack, n, rotate, d_out_sv, stream);
} PLSX_CATCH(ctx)

namespace {
// Dual permutation path.  A permutation leaves the feature side untouched
// (behavioral PLS permutes Y, base.py:599; mean-centred PLS permutes the rows
// of X but applies no per-feature scaling, meancentered.py:125), so its
// cross-covariance is R_p = A_p . Xf with ONE fixed feature matrix Xf (the
// cell-z-scored X, or the centred X for covariance / mean-centred PLS) and the
// permutation statistic -- singular values of R_p, optionally Procrustes-rotated
// on the T' side (base.py:683-712) -- needs only the Gram matrix
//     G_p = R_p R_p^T = A_p (Xf Xf^T) A_p^T = A_p K A_p^T,   K = Xf Xf^T  (S x S).
// K is formed once per call (one pass over X); every permutation then costs
// O(T' S^2) instead of O(T' S B).
int perm_dual(plsx_ctx* ctx, const int32_t* d_perm_idx, const double* d_ystack, int n, int rotate,
              double* d_out_sv, hipStream_t st)
{
    const int S = ctx->S, Tp = ctx->Tp, Sd = round_up(S, 8);
    // K depends on the bound data only: formed by the first permutation call after
    // plsx_set_data / plsx_set_perm_path and kept for the later ones (a front-end that
    // ships its permutations in chunks as the index rows arrive pays one pass over X)
    if (!ctx->has_Kd) {
        if (int e = ensure(ctx, ctx->Kd, (size_t)S * Sd * 8, true)) return e;
        const double* Xf = ctx->has_Xn ? ptr<double>(ctx->Xn) : ptr<double>(ctx->Xc);
        if (int e = run_nt(ctx, Xf, 0, ctx->Bpad, S, Xf, 0, ctx->Bpad, S, nullptr, 0, 0, 0, ctx->B, 1,
                           ptr<double>(ctx->Kd), 0, Sd, nullptr, 0, 0, st, true))
            return e;
        ctx->has_Kd = 1;
    }
    // resamples per pass: 2 GB operands, grid.y / grid.z limits of the tiled GEMM
    long long nb = std::min<long long>(32768, (2LL << 30) / ((long long)Tp * Sd * 8));
    nb = std::min<long long>(nb, 60000LL * 64 / ((long long)Tp * ceil_div(S, 64)));
    nb = std::max<long long>(nb, 1);
    GroupLayout lay;
    memset(&lay, 0, sizeof(lay));
    lay.n = 1; lay.Tp = Tp; lay.J = ctx->J; lay.T = ctx->T; lay.MT = ctx->MT; lay.Tpp = ctx->Tpp;
    for (int off = 0; off < n; off += (int)nb) {
        const int m = std::min<int>((int)nb, n - off);
        const size_t abytes = (size_t)m * Tp * Sd * 8;
        if (int e = ensure(ctx, ctx->Ad, abytes)) return e;
        if (int e = ensure(ctx, ctx->Wd, abytes)) return e;
        if (int e = ensure(ctx, ctx->Gm, (size_t)m * Tp * Tp * 8)) return e;
        HIPCHK(hipMemsetAsync(ctx->Ad.p, 0, abytes, st));
        const int* idx = d_perm_idx ? d_perm_idx + (size_t)off * S : nullptr;
        if (ctx->method == PLSX_BEHAVIORAL) {
            const double* yst = d_ystack ? d_ystack + (size_t)off * S * ctx->T : nullptr;
            hipLaunchKernelGGL(k_build_A_behav, dim3(m, ctx->J), dim3(256), (size_t)2 * ctx->T * 8, st,
                               yst ? yst : ptr<double>(ctx->Y), yst ? (long long)S * ctx->T : 0LL, ctx->T, S,
                               ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), (const int*)nullptr, idx,
                               lay, ctx->cov, 0, ptr<double>(ctx->Ad), (size_t)0, (double*)nullptr, 0, Sd);
        } else {
            hipLaunchKernelGGL(k_build_A_mc, dim3(m), dim3(256), 0, st, S, ctx->J, ctx->n_cond, ctx->mc,
                               ptr<int>(ctx->cell_of_row), idx, lay, ptr<double>(ctx->Ad), (size_t)0, Sd);
        }
        LAUNCHCHK();
        // W = A K  (all permutations stacked: (m T') x S)
        if (int e = run_nt(ctx, ptr<double>(ctx->Ad), 0, Sd, m * Tp, ptr<double>(ctx->Kd), 0, Sd, S,
                           nullptr, 0, 0, 0, S, 1, ptr<double>(ctx->Wd), 0, Sd, nullptr, 0, 0, st))
            return e;
        // G_p = W_p A_p^T
        if (int e = run_dual_gp(ctx, m, Sd, nullptr, 0, st)) return e;
        SmallArgs a = small_args(ctx, SMALL_PERM);
        a.rotate = rotate ? 1 : 0;
        a.out_sv = d_out_sv + (size_t)off * ctx->L;
        if (int e = run_small(ctx, a, m, st)) return e;
    }
    return PLSX_OK;
}

int perm_batch_impl(plsx_ctx* ctx, const int32_t* d_perm_idx, const double* d_ystack, int n, int rotate,
                    double* d_out_sv, void* stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    if (use_dual(ctx)) return perm_dual(ctx, d_perm_idx, d_ystack, n, rotate, d_out_sv, st);
    const int pg = ctx->fix ? ctx->npgf : ctx->npg;
    const int nb = balanced_batch(n, launch_groups(ctx, n, pg) * pg, pg);
    for (int off = 0; off < n; off += nb) {
        const int m = std::min(nb, n - off);
        const int* idx = d_perm_idx ? d_perm_idx + (size_t)off * ctx->S : nullptr;
        // behavioral permutes Y (base.py:599), mean-centred permutes X (meancentered.py:125)
        const int* xs = (ctx->method == PLSX_BEHAVIORAL) ? nullptr : idx;
        const int* ys = (ctx->method == PLSX_BEHAVIORAL) ? idx : nullptr;
        const double* yst = d_ystack ? d_ystack + (size_t)off * ctx->S * ctx->T : nullptr;
        if (ctx->fix) {
            if (int e = run_xprod_fixed(ctx, ys, m, st, yst)) return e;
        } else if (int e = run_xprod(ctx, xs, ys, m, st, false, yst)) return e;
        if (int e = run_gram(ctx, m, false, st)) return e;
        SmallArgs a = small_args(ctx, SMALL_PERM);
        a.rotate = rotate ? 1 : 0;
        a.out_sv = d_out_sv + (size_t)off * ctx->L;
        if (int e = run_small(ctx, a, m, st, ptr<double>(ctx->R))) return e;
    }
    return PLSX_OK;
}

}  // namespace

}  // extern "C"

namespace {
// ---- quadratic-form route of the bootstrap sums -------------------------------------------------
// Where the bootstrap weights are linear in a FIXED feature matrix, U_b = Xc^T V_b with V_b (S x L) known in dual
// space (single-pass bootstraps of the unscaled PLS-C modes; SIMPLS with the signs aligned in dual space),
//   sum_b U_b = Xc^T (sum_b V_b),     sum_b U_b[j,l]^2 = x_j^T C_l x_j,   C_l = sum_b v_bl v_bl^T  (S x S)
// and the pass over the B features happens ONCE per series of bootstraps (plsx_boot_finish: 2 S^2 L B flop) instead
// of once per bootstrap (2 S L B n flop): c3 (S = 200, 10 000 bootstraps) 50 x less matrix work, c5 (S = 1000,
// 5000) 5 x.  A series is announced with plsx_boot_begin(n); with n below ~S the per-bootstrap pass is cheaper and
// stays.  Same sums to rounding (every term of the direct sum is one term of the quadratic form, re-associated).
bool quad_applicable(const plsx_ctx* ctx);

// Per batch: V dense [m][L * S] (ctx->Vdq) -> transposed [L * S][mpad] -> C_l += Vt_l Vt_l^T, Vsum += row sums.
int quad_accumulate(plsx_ctx* ctx, int m, hipStream_t st)
{
    const int S = ctx->S, L = ctx->method == PLSX_REGRESSION ? ctx->ncomp : ctx->L;
    const int mpad = round_up(m, 2);
    const long long rows = (long long)L * S;
    if (int e = ensure(ctx, ctx->Vtq, (size_t)rows * mpad * 8)) return e;
    {
        KTimer tm(ctx, KC_BUILD, st);
        if (mpad != m) HIPCHK(hipMemsetAsync(ctx->Vtq.p, 0, (size_t)rows * mpad * 8, st));
        hipLaunchKernelGGL(k_transpose, dim3(ceil_div((int)rows, 32), ceil_div(m, 32)), dim3(32, 8), 0, st,
                           ptr<double>(ctx->Vdq), m, (int)rows, (int)rows, ptr<double>(ctx->Vtq), mpad);
        LAUNCHCHK();
        hipLaunchKernelGGL(k_rowsum_acc, dim3(ceil_div((int)rows, 4)), dim3(256), 0, st, ptr<double>(ctx->Vtq), mpad, m,
                           (int)rows, ptr<double>(ctx->Vsumq));
        LAUNCHCHK();
    }
    const double* Vt = ptr<double>(ctx->Vtq);
    if (int e = run_nt(ctx, Vt, (long long)S * mpad, mpad, S, Vt, (long long)S * mpad, mpad, S, nullptr, 0, 0, 0, m, L,
                       ptr<double>(ctx->Cq), (long long)S * S, S, nullptr, 0, 0, st, true, true))
        return e;
    ctx->quad_n += m;
    return 0;
}

inline int quad_blocks(int tiles) { return std::max(ceil_div(tiles, 24), tiles >= 8 ? 2 : 1); }

// The closing pass of a series: usum += Xc^T Vsum, usq[j][l] += x_j^T C_l x_j.
template <int MT>
int quad_finish_t(plsx_ctx* ctx, double* d_usq, int gpl, hipStream_t st)
{
    constexpr int KT = 1, NW = 8;
    const int S = ctx->S, L = ctx->method == PLSX_REGRESSION ? ctx->ncomp : ctx->L, B = ctx->B;
    const size_t gstride = (size_t)ctx->nks * MT * 64;
    ctx->quad_MT = MT; ctx->quad_gpl = gpl;
    const bool full = ctx->opt[OPT_QUAD_FULL_ROWS] != 0;      // A/B: every row block over all S columns (no use of the symmetry)
    if (ctx->timing) ++ctx->quad_series;
    // l's per pass: A operands within 1 GB
    const int lmax = (int)std::max<size_t>(1, std::min<size_t>((size_t)L, (1ULL << 30) / (gstride * 8 * gpl)));
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    HIPCHK(set_lds(k_xprod<MT, NW, KT, 0, 7>, stage));
    const int ncolblk = ceil_div(ctx->Bpad, NW * 16);
    for (int l0 = 0; l0 < L; l0 += lmax) {
        const int nl = std::min(lmax, L - l0), groups = nl * gpl;
        if (int e = ensure(ctx, ctx->Afrag_q, (size_t)groups * gstride * 8 + 4096)) return e;
        if (int e = ensure(ctx, ctx->qpart, (size_t)groups * ctx->Bpad * 8)) return e;
        HIPCHK(hipMemsetAsync(ctx->Afrag_q.p, 0, (size_t)groups * gstride * 8, st));
        {
            KTimer tm(ctx, KC_BUILD, st);
            hipLaunchKernelGGL(k_pack_afrag, dim3(64, groups), dim3(256), 0, st,
                               ptr<double>(ctx->Cq) + (size_t)l0 * S * S, S, gpl, MT, ptr<double>(ctx->Afrag_q), gstride, full ? 1 : 0);
            LAUNCHCHK();
        }
        SplitEpi se;
        memset(&se, 0, sizeof(se));
        se.acc_sum = ptr<double>(ctx->qpart); se.npairs = gpl; se.accB = S; se.nmu = full ? 1 : 0; se.J = nl;
        // Groups are numbered row block first: the groups of a sweep go to the eight XCDs in lockstep, and with the
        // blocks of an LV next to each other (contraction lengths S, 2 S / 3, S / 3 at c5) the XCDs with short blocks
        // waited for the one with the long block: 38.7 ms, block-major 35.1, one launch per row block (A/B option) 34.9
        const bool per_block = ctx->opt[OPT_QUAD_LAUNCH_PER_BLOCK] != 0;
        for (int pb = 0; pb < (per_block ? gpl : 1); ++pb) {
            const int ng = per_block ? nl : groups;
            se.Tpp = pb;
            se.acc_sum = ptr<double>(ctx->qpart) + (size_t)pb * nl * ctx->Bpad;
            KTimer tm(ctx, KC_XPROD, st);
            hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0, 7>), dim3(ncolblk * round_up(ng, 8)), dim3(NW * 64), stage, st,
                               ptr<double>(ctx->Afrag_q) + (size_t)pb * nl * gstride, gstride, ptr<double>(ctx->Xc), ctx->Bpad,
                               ctx->nks, (double*)nullptr, ctx->Bpad, 0, (const int*)nullptr, (const int*)nullptr,
                               (const double*)nullptr, 0, ng, ncolblk, (double*)nullptr, se, 1);
            LAUNCHCHK();
        }
        {
            KTimer tm(ctx, KC_UROT, st);
            // (usq is [B][L]: the pass of l0.. adds into columns l0..)
            hipLaunchKernelGGL(k_quad_finish, dim3((unsigned)(((long long)B * nl + 255) / 256)), dim3(256), 0, st,
                               ptr<double>(ctx->qpart), gpl, ctx->Bpad, B, nl, L, l0, d_usq);
            LAUNCHCHK();
        }
    }
    return 0;
}

int quad_finish(plsx_ctx* ctx, double* d_usum, double* d_usq, hipStream_t st)
{
    const int S = ctx->S, L = ctx->method == PLSX_REGRESSION ? ctx->ncomp : ctx->L, B = ctx->B;
    {
        KTimer tm(ctx, KC_UROT, st);
        hipLaunchKernelGGL(k_xt_vsum, dim3(ceil_div(B, 256), ceil_div(L, 8)), dim3(256), 0, st, ptr<double>(ctx->Xc),
                           ctx->Bpad, S, B, ptr<double>(ctx->Vsumq), L, d_usum);
        LAUNCHCHK();
    }
    // the S rows of a C_l in gpl blocks of MT tiles, as evenly as the instantiated block heights allow
    // (at least two blocks once there are 8 tiles: the second one starts its contraction half way down)
    const int tiles = ceil_div(S, 16), gpl = quad_blocks(tiles);
    const int need = ctx->opt[OPT_QUAD_MT] > 0 ? std::min(24, ctx->opt[OPT_QUAD_MT]) : ceil_div(tiles, gpl);
    if (need <= 8) return quad_finish_t<8>(ctx, d_usq, ceil_div(tiles, 8), st);
    if (need <= 12) return quad_finish_t<12>(ctx, d_usq, ceil_div(tiles, 12), st);
    if (need <= 16) return quad_finish_t<16>(ctx, d_usq, ceil_div(tiles, 16), st);
    if (need <= 20) return quad_finish_t<20>(ctx, d_usq, ceil_div(tiles, 20), st);
    if (need <= 21) return quad_finish_t<21>(ctx, d_usq, ceil_div(tiles, 21), st);
    if (need <= 22) return quad_finish_t<22>(ctx, d_usq, ceil_div(tiles, 22), st);
    return quad_finish_t<24>(ctx, d_usq, gpl, st);
}
}  // namespace

namespace {
// Single-pass bootstrap of the UNSCALED modes (mean-centred PLS, behavioral PLS in covariance
// mode).  Without per-feature scaling R_r = A_r Xc is linear in the fixed feature matrix, so
//   G_r = R_r R_r^T = A_r K A_r^T            (K = Xc Xc^T, S x S: the kernel of the dual permutation route)
//   P_r = R_r U0   = A_r (Xc U0) = A_r Sc    (Sc = the score columns appended to Xc; also = gen_distrib)
// need no pass over the features, and with M_r from the small solver
//   U_r = R_r^T M_r = Xc^T (A_r^T M_r) = Xc^T W_r
// is ONE cross-product pass whose epilogue adds U_r and U_r^2 over the resamples of a group
// (k_xprod EPI = 2): no R matrix is written, no Gram pass and no rotation pass read it back.
// Same statistics to rounding as the two-pass route (tests: test_single_pass_bootstrap_*).
int boot_single_pass(plsx_ctx* ctx, const int32_t* d_boot_idx, int n, double* d_usum, double* d_usq,
                     double* d_distrib, hipStream_t st)
{
    const int S = ctx->S, Tp = ctx->Tp, L = ctx->L, Sd = round_up(S, 8), MT = 24;
    const int npg_w = (MT * 16) / L;
    const size_t gstride = (size_t)ctx->nks * MT * 64;
    if (!ctx->has_Kd) {
        if (int e = ensure(ctx, ctx->Kd, (size_t)S * Sd * 8, true)) return e;
        const double* Xf = ptr<double>(ctx->Xc);
        if (int e = run_nt(ctx, Xf, 0, ctx->Bpad, S, Xf, 0, ctx->Bpad, S, nullptr, 0, 0, 0, ctx->B, 1,
                           ptr<double>(ctx->Kd), 0, Sd, nullptr, 0, 0, st, true))
            return e;
        ctx->has_Kd = 1;
    }
    if (ctx->npg_w != npg_w) {
        std::vector<int> lmap(MT * 16, -1);
        for (int rr = 0; rr < npg_w; ++rr)
            for (int l = 0; l < L; ++l) lmap[rr * L + l] = l;
        if (int e = ensure(ctx, ctx->out_row_w, lmap.size() * sizeof(int))) return e;
        HIPCHK(hipMemcpy(ctx->out_row_w.p, lmap.data(), lmap.size() * sizeof(int), hipMemcpyHostToDevice));
        ctx->npg_w = npg_w;
    }
    // resamples per pass: partial (sum, sum of squares) tiles of every group [groups][B][L] x 2 within a
    // quarter of the scratch budget, dense operands within 2 GB, grid limits of the tiled GEMM
    const double per_group = 2.0 * ctx->B * (double)L * 8.0;
    long long gmax = (long long)(ctx->scratch_gb * 1073741824.0 / 4.0 / per_group);
    gmax = std::max<long long>(1, std::min<long long>(gmax, 512));
    long long nb = gmax * npg_w;
    nb = std::min<long long>(nb, (2LL << 30) / ((long long)Tp * Sd * 8));
    nb = std::min<long long>(nb, 60000LL * 64 / ((long long)Tp * ceil_div(S, 64)));
    if (ctx->quad_active)          // V of a batch, dense and transposed, within 1 GB each
        nb = std::min<long long>(nb, std::max<long long>(npg_w, (1LL << 30) / ((long long)L * S * 8)));
    nb = std::max<long long>(npg_w, (nb / npg_w) * npg_w);
    GroupLayout lay;
    memset(&lay, 0, sizeof(lay));
    lay.n = 1; lay.Tp = Tp; lay.J = ctx->J; lay.T = ctx->T; lay.MT = ctx->MT; lay.Tpp = ctx->Tpp;
    const size_t mstride = (size_t)ctx->nks_t * ctx->LT * 64;
    for (int off = 0; off < n; off += (int)nb) {
        const int m = std::min<int>((int)nb, n - off);
        const int groups = ceil_div(m, npg_w);
        const size_t abytes = (size_t)m * Tp * Sd * 8;
        if (int e = ensure(ctx, ctx->Ad, abytes)) return e;
        if (int e = ensure(ctx, ctx->Wd, abytes)) return e;
        if (int e = ensure(ctx, ctx->Gm, (size_t)m * Tp * Tp * 8)) return e;
        if (int e = ensure(ctx, ctx->Pm, (size_t)m * Tp * L * 8)) return e;
        if (int e = ensure(ctx, ctx->Mfrag, (size_t)m * mstride * 8 + 1024)) return e;
        if (!ctx->quad_active) {
            if (int e = ensure(ctx, ctx->Afrag, (size_t)groups * gstride * 8 + 4096)) return e;
            if (int e = ensure(ctx, ctx->psum, (size_t)groups * ctx->B * L * 8)) return e;
            if (int e = ensure(ctx, ctx->psq, (size_t)groups * ctx->B * L * 8)) return e;
        }
        if (ctx->timing) ctx->timed_units += m;
        HIPCHK(hipMemsetAsync(ctx->Ad.p, 0, abytes, st));
        const int* idx = d_boot_idx + (size_t)off * S;
        {
            KTimer tm(ctx, KC_BUILD, st);
            if (ctx->method == PLSX_BEHAVIORAL)
                hipLaunchKernelGGL(k_build_A_behav, dim3(m, ctx->J), dim3(256), (size_t)2 * ctx->T * 8, st,
                                   ptr<double>(ctx->Y), 0LL, ctx->T, S, ptr<int>(ctx->cell_start),
                                   ptr<int>(ctx->cell_len), idx, idx, lay, ctx->cov, 0, ptr<double>(ctx->Ad),
                                   (size_t)0, (double*)nullptr, 0, Sd);
            else
                hipLaunchKernelGGL(k_build_A_mc, dim3(m), dim3(256), 0, st, S, ctx->J, ctx->n_cond, ctx->mc,
                                   ptr<int>(ctx->cell_of_row), idx, lay, ptr<double>(ctx->Ad), (size_t)0, Sd);
            LAUNCHCHK();
        }
        // W = A K (all resamples stacked), G_r = W_r A_r^T, P_r = A_r Sc
        if (int e = run_nt(ctx, ptr<double>(ctx->Ad), 0, Sd, m * Tp, ptr<double>(ctx->Kd), 0, Sd, S,
                           nullptr, 0, 0, 0, S, 1, ptr<double>(ctx->Wd), 0, Sd, nullptr, 0, 0, st))
            return e;
        if (int e = run_dual_gp(ctx, m, Sd, ptr<double>(ctx->ScT), L, st)) return e;
        SmallArgs a = small_args(ctx, SMALL_BOOT);
        if (int e = run_small(ctx, a, m, st)) return e;
        // gen_distrib of a resample is its cross-product with the score columns: P_r itself
        HIPCHK(hipMemcpyAsync(d_distrib + (size_t)off * Tp * L, ctx->Pm.p, (size_t)m * Tp * L * 8,
                              hipMemcpyDeviceToDevice, st));
        if (ctx->quad_active) {
            // quadratic-form route: W_r stays in dual space; the feature pass comes once, in plsx_boot_finish
            if (int e = ensure(ctx, ctx->Vdq, (size_t)m * L * S * 8)) return e;
            {
                KTimer tm(ctx, KC_BUILD, st);
                hipLaunchKernelGGL(k_build_Vd, dim3(m), dim3(256), (size_t)Tp * L * 8, st, ptr<double>(ctx->Ad), Sd, S, Tp, L,
                                   ptr<double>(ctx->Mfrag), ctx->nks_t, ctx->LT, ptr<double>(ctx->Vdq));
                LAUNCHCHK();
            }
            if (int e = quad_accumulate(ctx, m, st)) return e;
            continue;
        }
        HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)groups * gstride * 8, st));
        {
            KTimer tm(ctx, KC_BUILD, st);
            hipLaunchKernelGGL(k_build_W, dim3(m), dim3(256), (size_t)Tp * L * 8, st, ptr<double>(ctx->Ad), Sd, S, Tp, L,
                               ptr<double>(ctx->Mfrag), ctx->nks_t, ctx->LT, npg_w, MT, ptr<double>(ctx->Afrag), gstride);
            LAUNCHCHK();
        }
        if (int e = launch_xprod_acc(ctx, ptr<double>(ctx->Afrag), gstride, groups, L, st)) return e;
        {
            KTimer tm(ctx, KC_UROT, st);          // the fixed-order sum over groups (what k_urot's splits do)
            const long long count = (long long)ctx->B * L;
            hipLaunchKernelGGL(k_add_splits, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                               ptr<double>(ctx->psum), ptr<double>(ctx->psq), groups, count, d_usum, d_usq);
            LAUNCHCHK();
        }
    }
    return PLSX_OK;
}
}  // namespace

extern "C" {

int plsx_boot_batch(plsx_ctx* ctx, const int32_t* d_boot_idx, int n, double* d_usum, double* d_usq,
                    double* d_distrib, void* stream)
try {
    NEED_ORIG();
    if (!d_boot_idx || !d_usum || !d_usq || !d_distrib || n < 1)
        return fail(ctx, PLSX_ERR_ARG, "plsx_boot_batch: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    // unscaled modes: one pass over the features per bootstrap (see boot_single_pass)
    if (plsc_single_pass(ctx))
        return boot_single_pass(ctx, d_boot_idx, n, d_usum, d_usq, d_distrib, st);
    if (ctx->quad_active) return fail(ctx, PLSX_ERR_STATE, "plsx_boot_batch: open series on a route that left it");
    const int nb = balanced_batch(n, launch_groups(ctx, n, ctx->npg) * ctx->npg, ctx->npg);
    for (int off = 0; off < n; off += nb) {
        const int m = std::min(nb, n - off);
        const int* idx = d_boot_idx + (size_t)off * ctx->S;
        if (int e = run_xprod(ctx, idx, idx, m, st, false, nullptr, -1, true)) return e;
        if (int e = run_gram(ctx, m, true, st)) return e;
        const double* R = ptr<double>(ctx->R);
        SmallArgs a = small_args(ctx, SMALL_BOOT);
        if (int e = run_small(ctx, a, m, st, R)) return e;
        if (int e = run_urot(ctx, m, d_usum, d_usq, nullptr, st)) return e;
        dim3 g(ceil_div(ctx->Tp * ctx->L, 256), m);
        hipLaunchKernelGGL(k_gather_cols, g, dim3(256), 0, st, R, ctx->strideR, ctx->Bpad, ctx->B, ctx->Tp,
                           ctx->L, d_distrib + (size_t)off * ctx->Tp * ctx->L);
        LAUNCHCHK();
    }
    return PLSX_OK;
} PLSX_CATCH(ctx)

}  // extern "C"

namespace {
// First halves of m splits through the cross-product kernel as raw sums; its
// epilogue writes both halves of split i to R slots 2i and 2i + 1 (see SplitEpi).
template <int NSQ>
int launch_xprod_split(plsx_ctx* ctx, int groups, SplitEpi se, hipStream_t st)
{
    constexpr int MT = 24, NW = 4, KT = 1;
    // LDS (doubles): region 0 = the two A stages, reused by the epilogue for its per-wave
    // column tables [NW][5][nmu][16] and the row maps; then, when it fits next to a second
    // resident block, the DMA-prefetched tile of Rfull [Tpp][64] and the row constants.
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128;
    se.nmu = ctx->npg * ctx->J;
    const size_t epi0 = (size_t)NW * 5 * se.nmu * 16 + (size_t)MT * 16;
    const size_t pre0 = round_up((int)std::max(stage, epi0), 128);
    const size_t pre_total = pre0 + (size_t)ctx->Tpp * NW * 16 + (size_t)MT * 16 * 5;
    se.off_pre = (pre_total * 8 <= 80 * 1024) ? (int)pre0 : 0;
    const size_t lds = se.off_pre ? pre_total * 8 : std::max(stage, epi0 + (size_t)MT * 16 * 5) * 8;
    HIPCHK(set_lds(k_xprod<MT, NW, KT, NSQ, true>, lds));
    const int ncolblk = ctx->Bpad / (NW * 16);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod<MT, NW, KT, NSQ, true>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), lds,
                       st, ptr<double>(ctx->Afrag), ctx->group_stride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                       ptr<double>(ctx->R), ctx->Bpad, ctx->npg * 2 * ctx->Tpp, ptr<int>(ctx->out_row_s),
                       ptr<int>(ctx->mom_idx), ptr<double>(ctx->mom_n), std::max(ctx->nmom_pad, 0), groups,
                       ncolblk, (double*)nullptr, se, 1);
    LAUNCHCHK();
    return 0;
}

// Compact fused split-half (T' <= 64): ONE split per cross-product block, contracting over the rows of its
// first half only (the fused epilogue derives the second half from the full-sample cross-product, so the
// zeros that the 7-splits-per-block layout multiplies for the other half are half of its MFMA work: seven
// random halves cover every subject between them, a block of its own covers S / 2).  Data blocks of
// ceil(T'/16) tiles with KT k-steps per LDS stage (24 tile-steps per barrier, as in the big blocks), X rows
// through a per-split row table (k_xprod IDX), first-half feature moments from moment-only blocks over all
// (split, cell) pairs of the pass (full K: 16 tile-steps per split).  432 -> 272 tile-steps per split at
// the headline shape; the price is one pass over half of X per split (0.4 GB): the leg turns HBM bound.
template <int MT, int KT, bool TAIL = false>
int launch_xprod_compact(plsx_ctx* ctx, int m, int nks_c, SplitEpi se, hipStream_t st)
{
    const int J = ctx->J;
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8 + (size_t)nks_c * 4 * sizeof(int);
    const size_t epi = (size_t)5 * J * 128 * 8 + (size_t)2 * MT * 16 * 4 + (size_t)MT * 16 * 5 * 8;
    const size_t lds = std::max(stage, epi);
    se.off_pre = 0;
    HIPCHK(set_lds(k_xprod_compact<MT, KT, 5, TAIL>, lds));
    const int ncolblk = ceil_div(ctx->Bpad, 128);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod_compact<MT, KT, 5, TAIL>), dim3(round_up(ncolblk, 8) * round_up(m, 8)), dim3(256), lds, st,
                       ptr<double>(ctx->Afrag_c), (size_t)nks_c * MT * 64, ptr<double>(ctx->Xc), ctx->Bpad, nks_c,
                       ptr<double>(ctx->R), ctx->Bpad, 2 * ctx->Tpp, ptr<int>(ctx->out_row_c), ptr<int>(ctx->mom_idx_c),
                       ptr<double>(ctx->momn_m), m, ncolblk, se);
    LAUNCHCHK();
    return 0;
}

int run_split_compact(plsx_ctx* ctx, const int* perm, const uint8_t* masks, int m, const double* Rfull,
                      hipStream_t st, const double* Yarr)
{
    const int J = ctx->J, S = ctx->S, MTc = ceil_div(ctx->Tp, 16), KT = 12 / MTc, rows = MTc * 16;
    ctx->last_compact_n = 0;                            // (the row tables are about to hold this pass's splits)
    if (!ctx->has_cellS) {
        if (int e = ensure(ctx, ctx->cellS, (size_t)2 * J * ctx->Bpad * 8, true)) return e;
        hipLaunchKernelGGL(k_cell_moments, dim3(ceil_div(ctx->B, 256)), dim3(256), 0, st, ptr<double>(ctx->Xc),
                           ctx->Bpad, ctx->B, J, ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len),
                           ptr<double>(ctx->cellS), ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad);
        LAUNCHCHK();
        ctx->has_cellS = 2;                             // (the 7-per-block row map of run_split_fused is not uploaded)
    }
    if (int e = ensure_compact_maps(ctx)) return e;
    // the tables are sized for a first half of all S rows; a block contracts over its own count
    const int nks_c = round_up(ceil_div(S, 4), KT);
    const size_t astride = (size_t)nks_c * MTc * 64;
    const int npairs = m * J;
    const MomLayout ml = moment_layout(ctx, npairs);
    const int groups_m = ml.groups;
    const size_t mstride = ml.stride;
    if (int e = ensure_scratch(ctx, std::min(ctx->Gcap, ceil_div(2 * m, ctx->npg)))) return e;
    if ((size_t)2 * m * ctx->strideR * 8 > ctx->R.bytes) return fail(ctx, PLSX_ERR_STATE, "compact split: R scratch too small");
    if (int e = ensure(ctx, ctx->Afrag_c, (size_t)m * astride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->rank_c, (size_t)m * S * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->rowtab_c, ((size_t)m * nks_c * 4 + m) * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->Afrag_m, (size_t)groups_m * mstride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->momn_m, (size_t)round_up(npairs, 192) * 8)) return e;
    if (int e = ensure(ctx, ctx->m1_c, (size_t)round_up(npairs, 8) * ctx->Bpad * 8)) return e;
    if (int e = ensure(ctx, ctx->m2_c, (size_t)round_up(npairs, 8) * ctx->Bpad * 8)) return e;
    if (int e = ensure(ctx, ctx->rowc, ((size_t)m * rows * 5 + 256) * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->Afrag_c.p, 0, (size_t)m * astride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->Afrag_m.p, 0, (size_t)groups_m * mstride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->rowc.p, 0, ((size_t)m * rows * 5 + 256) * 8, st));
    {
        KTimer tm(ctx, KC_BUILD, st);
        hipLaunchKernelGGL(k_split_rank, dim3(m), dim3(64), 0, st, masks, S, nks_c * 4, ptr<int>(ctx->rank_c),
                           ptr<int>(ctx->rowtab_c), ptr<int>(ctx->rowtab_c) + (size_t)m * nks_c * 4);
        LAUNCHCHK();
        GroupLayout lay;
        lay.n = 1; lay.Tp = ctx->Tp; lay.J = J; lay.T = ctx->T; lay.MT = MTc; lay.w0 = MTc; lay.sq0 = MTc; lay.Tpp = ctx->Tpp;
        hipLaunchKernelGGL(k_build_A_split, dim3(m, J), dim3(256), 0, st, Yarr ? Yarr : ptr<double>(ctx->Y), ctx->T, S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), perm, masks, lay,
                           ptr<double>(ctx->Afrag_c), astride, ptr<double>(ctx->momn_m), 0, ptr<double>(ctx->rowc),
                           ptr<int>(ctx->rank_c), ptr<double>(ctx->Afrag_m), mstride, ml.pairs);
        LAUNCHCHK();
    }
    SplitEpi se;
    memset(&se, 0, sizeof(se));
    se.scale = ptr<double>(ctx->m1_c); se.scale2 = ptr<double>(ctx->m2_c);
    se.npairs = npairs;
    if (int e = launch_moment_blocks<6>(ctx, ml, se, st)) return e;     // (4-wave blocks: two raw tables to write)
    se.Rfull = Rfull;
    se.cellS1 = ptr<double>(ctx->cellS);
    se.cellS2 = ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad;
    se.cell_len = ptr<int>(ctx->cell_len);
    se.rowc = ptr<double>(ctx->rowc);
    se.J = J; se.Tpp = ctx->Tpp;
    se.row_tab = ptr<int>(ctx->rowtab_c);
    se.row_cnt = ptr<int>(ctx->rowtab_c) + (size_t)m * nks_c * 4;
    // (a last tile of <= 4 live rows -- T' = 50: rows 48, 49 -- runs on the 4x4x4 shape)
    const bool tail = ctx->Tp - (MTc - 1) * 16 <= 4 && !ctx->opt[OPT_SPLIT_NO_TAIL4];
    switch (MTc) {
        case 1: return launch_xprod_compact<1, 12>(ctx, m, nks_c, se, st);
        case 2: return tail ? launch_xprod_compact<2, 6, true>(ctx, m, nks_c, se, st)
                            : launch_xprod_compact<2, 6>(ctx, m, nks_c, se, st);
        case 3: return tail ? launch_xprod_compact<3, 4, true>(ctx, m, nks_c, se, st)
                            : launch_xprod_compact<3, 4>(ctx, m, nks_c, se, st);
        default: return tail ? launch_xprod_compact<4, 3, true>(ctx, m, nks_c, se, st)
                             : launch_xprod_compact<4, 3>(ctx, m, nks_c, se, st);
    }
}

int run_split_fused(plsx_ctx* ctx, const int* perm, const uint8_t* masks, int m, const double* Rfull,
                    hipStream_t st, const double* Yarr)
{
    // (LDS of a compact block: the row table, 4 S bytes; the epilogue's five column tables of every cell, 5 KB each)
    if (ctx->Tp <= 64 && ctx->J <= 10 && ctx->S <= 8192 && (long long)ctx->Kpad * ctx->Bpad * 8 < (1LL << 31) &&
        !ctx->opt[OPT_SPLIT_INBLOCK])
        return run_split_compact(ctx, perm, masks, m, Rfull, st, Yarr);
    const int J = ctx->J, S = ctx->S, rows = ctx->MT * 16;
    if (ctx->has_cellS != 1) {
        if (int e = ensure(ctx, ctx->cellS, (size_t)2 * J * ctx->Bpad * 8, true)) return e;
        hipLaunchKernelGGL(k_cell_moments, dim3(ceil_div(ctx->B, 256)), dim3(256), 0, st, ptr<double>(ctx->Xc),
                           ctx->Bpad, ctx->B, J, ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len),
                           ptr<double>(ctx->cellS), ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad);
        LAUNCHCHK();
        std::vector<int> orow(rows, -1);
        for (int rr = 0; rr < ctx->npg; ++rr)
            for (int t = 0; t < ctx->Tp; ++t) orow[rr * ctx->Tp + t] = rr * 2 * ctx->Tpp + t;
        if (int e = ensure(ctx, ctx->out_row_s, rows * sizeof(int))) return e;
        HIPCHK(hipMemcpy(ctx->out_row_s.p, orow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
        ctx->has_cellS = 1;
    }
    const int groups = ceil_div(m, ctx->npg);
    if (int e = ensure_scratch(ctx, 2 * groups)) return e;
    if (int e = ensure(ctx, ctx->rowc, (size_t)groups * rows * 5 * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)groups * ctx->group_stride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->mom_n.p, 0, (size_t)groups * std::max(ctx->nmom_pad, 16) * 8, st));
    HIPCHK(hipMemsetAsync(ctx->rowc.p, 0, (size_t)groups * rows * 5 * 8, st));
    GroupLayout lay;
    lay.n = ctx->npg; lay.Tp = ctx->Tp; lay.J = J; lay.T = ctx->T; lay.MT = ctx->MT;
    lay.w0 = ctx->w0; lay.sq0 = ctx->sq0; lay.Tpp = ctx->Tpp;
    hipLaunchKernelGGL(k_build_A_split, dim3(m, J), dim3(256), 0, st, Yarr ? Yarr : ptr<double>(ctx->Y), ctx->T, S,
                       ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), perm, masks, lay,
                       ptr<double>(ctx->Afrag), ctx->group_stride, ptr<double>(ctx->mom_n), ctx->nmom_pad,
                       ptr<double>(ctx->rowc));
    LAUNCHCHK();
    SplitEpi se;
    se.Rfull = Rfull;
    se.cellS1 = ptr<double>(ctx->cellS);
    se.cellS2 = ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad;
    se.cell_len = ptr<int>(ctx->cell_len);
    se.rowc = ptr<double>(ctx->rowc);
    se.J = J; se.Tpp = ctx->Tpp;
    switch (ctx->MT - ctx->sq0) {
        case 1: return launch_xprod_split<1>(ctx, groups, se, st);
        case 2: return launch_xprod_split<2>(ctx, groups, se, st);
        default: return launch_xprod_split<3>(ctx, groups, se, st);
    }
}
}  // namespace

extern "C" {

int plsx_split_half_batch(plsx_ctx* ctx, const int32_t* d_perm_idx, int np, const uint8_t* d_masks,
                          int ns, double* d_ucorr, double* d_vcorr, void* stream)
try {
    return plsx_split_half_batch_y(ctx, d_perm_idx, nullptr, np, d_masks, ns, d_ucorr, d_vcorr, stream);
} PLSX_CATCH(ctx)

int plsx_split_half_batch_y(plsx_ctx* ctx, const int32_t* d_perm_idx, const double* d_ystack, int np,
                            const uint8_t* d_masks, int ns, double* d_ucorr, double* d_vcorr, void* stream)
try {
    NEED_DATA();
    if (!d_masks || !d_ucorr || !d_vcorr || np < 1 || ns < 1 || (d_ystack && d_perm_idx))
        return fail(ctx, PLSX_ERR_ARG, "plsx_split_half_batch: bad arguments");
    if (d_ystack && ctx->method != PLSX_BEHAVIORAL)
        return fail(ctx, PLSX_ERR_ARG, "plsx_split_half_batch_y: pre-permuted Y stacks need behavioral PLS");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    const int S = ctx->S, Tp = ctx->Tp, L = ctx->L;
    const int nb = ((launch_groups(ctx, 2LL * np * ns, ctx->npg) * ctx->npg) / 2) * 2;   // slots per super-batch (pairs of halves)
    if (nb < 2) return fail(ctx, PLSX_ERR_UNSUPPORTED, "plsx_split_half_batch: scratch too small");
    // Arrangements (permutations) are decomposed in chunks: one cross-product launch, one
    // Gram launch and one small-solver launch for up to `pcmax` of them instead of three
    // latency-bound launches per permutation (4 ms each at c4, as much as ten of its splits).
    const size_t mstride = (size_t)ctx->nks_t * ctx->LT * 64;
    int pcmax = 1;
    if (d_perm_idx || d_ystack) {
        const long long by_mem = std::max<long long>(1, (4LL << 30) / (ctx->strideR * 8));
        pcmax = (int)std::min<long long>(std::min<long long>(64, by_mem), np);
        pcmax = std::min(pcmax, std::max(1, ctx->Gcap * ctx->npg));
    }
    if (int e = ensure(ctx, ctx->Rfull, (size_t)pcmax * ctx->strideR * 8)) return e;
    if (int e = ensure(ctx, ctx->Vp, (size_t)pcmax * Tp * L * 8)) return e;
    if (int e = ensure(ctx, ctx->dp, (size_t)pcmax * L * 8)) return e;
    if (int e = ensure(ctx, ctx->Mvd, (size_t)pcmax * mstride * 8 + 1024)) return e;
    const int permute_x = (ctx->method == PLSX_MEANCENTERED) ? 1 : 0;
    // behavioral correlation mode: only the first half of a split takes the MFMA pass
    const bool fused = ctx->scaled && !permute_x && ctx->gps == 0 && ctx->Gcap >= 2 && ctx->MT == 24 &&
                       !ctx->opt[OPT_NO_SPLIT_FUSE];       // (the fused epilogue is instantiated for 24-tile blocks)
    // splits per pass (the fused path writes two R slots per split from groups of npg splits)
    const int spp = fused ? std::max(1, std::min(nb / 2, (ctx->Gcap / 2) * ctx->npg)) : nb / 2;
    for (int p0 = 0; p0 < np; p0 += pcmax) {
        const int pc = std::min(pcmax, np - p0);
        const int* pblock = d_perm_idx ? d_perm_idx + (size_t)p0 * S : nullptr;
        const size_t ysz = (size_t)S * ctx->T;
        // full-sample arrangements: R_p, then V_p, d_p and M = V_p / d_p (= vd, fragment order)
        if (int e = run_xprod(ctx, permute_x ? pblock : nullptr, permute_x ? nullptr : pblock, pc, st, false,
                              d_ystack ? d_ystack + (size_t)p0 * ysz : nullptr))
            return e;
        HIPCHK(hipMemcpyAsync(ctx->Rfull.p, ctx->R.p, (size_t)pc * ctx->strideR * 8, hipMemcpyDeviceToDevice, st));
        if (int e = run_gram(ctx, pc, false, st)) return e;
        SmallArgs a = small_args(ctx, SMALL_DECOMP);
        a.out_V = ptr<double>(ctx->Vp); a.out_d = ptr<double>(ctx->dp); a.Mfrag = ptr<double>(ctx->Mvd);
        if (int e = run_small(ctx, a, pc, st, ptr<double>(ctx->R))) return e;
        for (int pi = 0; pi < pc; ++pi) {
            const int p = p0 + pi;
            const int* perm = d_perm_idx ? d_perm_idx + (size_t)p * S : nullptr;
            const double* Rfull = ptr<double>(ctx->Rfull) + (size_t)pi * ctx->strideR;
            const double* Vp = ptr<double>(ctx->Vp) + (size_t)pi * Tp * L;
            const double* dp = ptr<double>(ctx->dp) + (size_t)pi * L;
            const double* Mvd = ptr<double>(ctx->Mvd) + (size_t)pi * mstride;
            const double* Yarr = d_ystack ? d_ystack + (size_t)p * ysz : nullptr;     // this arrangement's Y
            for (int off = 0; off < ns; off += spp) {
                const int m = std::min(spp, ns - off);               // splits in this pass
                if (fused) {
                    if (int e = run_split_fused(ctx, perm, d_masks + ((size_t)p * ns + off) * S, m, Rfull, st, Yarr))
                        return e;
                } else {
                    if (int e = ensure(ctx, ctx->srcx, (size_t)2 * m * S * sizeof(int))) return e;
                    if (int e = ensure(ctx, ctx->srcy, (size_t)2 * m * S * sizeof(int))) return e;
                    hipLaunchKernelGGL(k_split_src, dim3(ceil_div(S, 256), 2 * m), dim3(256), 0, st, perm,
                                       d_masks + ((size_t)p * ns + off) * S, m, S, permute_x,
                                       ptr<int>(ctx->srcx), ptr<int>(ctx->srcy));
                    LAUNCHCHK();
                    if (int e = run_xprod(ctx, ptr<int>(ctx->srcx), permute_x ? nullptr : ptr<int>(ctx->srcy),
                                          2 * m, st, false, Yarr, 0))
                        return e;
                }
                // C_h = D_h . R_p^T  (T' x T')
                if (int e = ensure(ctx, ctx->Cm, (size_t)2 * m * Tp * Tp * 8)) return e;
                if (int e = run_gram_ex(ctx, 2 * m, 2, Rfull, Tp, ptr<double>(ctx->Cm), st)) return e;
                // feature-axis sums of E_h = D_h^T . vd
                const int ntile = ceil_div(ctx->B, 16);
                int nchunk = std::min(std::max(1, ceil_div(2048, m)), std::max(1, ntile / 8));
                {
                    // whole rounds of resident blocks: 100 splits x 21 chunks = 4.1 rounds of 512 left the chip
                    // nearly empty for a fifth of the kernel
                    static const int slots = chip_slots(reinterpret_cast<const void*>(k_ucorr_partial<4, 13, true>));
                    nchunk = pick_parts(m, slots, std::max(1, (nchunk * 2) / 3), std::min(std::max(1, ntile / 8), 2 * nchunk));
                }
                const int tpc = ceil_div(ntile, nchunk);
                nchunk = ceil_div(ntile, tpc);
                const int lpad = ctx->LT * 16;
                if (int e = ensure(ctx, ctx->part2, (size_t)nchunk * m * 5 * lpad * 8)) return e;
                dim3 grid(nchunk, m), block(256);
                if (int e = launch_ucorr(ctx, grid, block, st, Mvd, tpc, ptr<double>(ctx->part2), m)) return e;
                LAUNCHCHK();
                hipLaunchKernelGGL(k_split_final, dim3(m), dim3(256), (size_t)4 * 256 * 5 * 8, st,
                                   ptr<double>(ctx->part2), nchunk, m,
                                   lpad, ptr<double>(ctx->Cm), Vp, dp, Tp, L, ctx->B,
                                   d_ucorr + ((size_t)p * ns + off) * L, d_vcorr + ((size_t)p * ns + off) * L);
                LAUNCHCHK();
            }
        }
    }
    return PLSX_OK;
} PLSX_CATCH(ctx)

}  // extern "C"

namespace {
int crossval_impl(plsx_ctx* ctx, const uint8_t* d_masks, int m, double* d_r, double* d_r2, hipStream_t st);

// Lay the groups out again with / without the per-cell moment rows (covariance mode carries
// them only while cross-validation needs the training mean / std of every feature).
int replan_moments(plsx_ctx* ctx, int want)
{
    ctx->cv_mom = want;
    if (plan_groups(ctx) != 0) return fail(ctx, PLSX_ERR_UNSUPPORTED, "cannot lay out the moment rows of a resample");
    ctx->Galloc = 0;                                   // scratch sizes follow the new group layout
    return upload_rowmaps(ctx);
}
}  // namespace

extern "C" {

int plsx_crossval_batch(plsx_ctx* ctx, const uint8_t* d_masks, int m, double* d_r, double* d_r2, void* stream)
try {
    NEED_DATA();
    if (ctx->method != PLSX_BEHAVIORAL)
        return fail(ctx, PLSX_ERR_ARG, "plsx_crossval_batch: cross-validation is defined for behavioral PLS");
    if (!d_masks || !d_r || !d_r2 || m < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_crossval_batch: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->momrows) return crossval_impl(ctx, d_masks, m, d_r, d_r2, st);
    HIPCHK(hipStreamSynchronize(st));                  // the row maps of queued launches are about to change
    if (int e = replan_moments(ctx, 1)) return e;
    const int rc = crossval_impl(ctx, d_masks, m, d_r, d_r2, st);
    HIPCHK(hipStreamSynchronize(st));
    if (int e = replan_moments(ctx, 0)) return e;
    return rc;
} PLSX_CATCH(ctx)

}  // extern "C"

namespace {
int crossval_impl(plsx_ctx* ctx, const uint8_t* d_masks, int m, double* d_r, double* d_r2, hipStream_t st)
{
    const int S = ctx->S, T = ctx->T, J = ctx->J, Tp = ctx->Tp, L = ctx->L;
    // splits per pass: bounded by the super-batch and by the J rescaled copies kept in R2
    int nb = std::max(1, std::min(launch_groups(ctx, m, ctx->npg) * ctx->npg, 256 / std::max(J, 1)));
    nb = (nb / ctx->npg) * ctx->npg;
    if (nb < ctx->npg) nb = ctx->npg;
    for (int off = 0; off < m; off += nb) {
        const int mm = std::min(nb, m - off);
        const int groups = ceil_div(mm, ctx->npg);
        const uint8_t* mk = d_masks + (size_t)off * S;
        if (int e = ensure(ctx, ctx->srcx, (size_t)mm * S * sizeof(int))) return e;
        hipLaunchKernelGGL(k_cv_src, dim3(ceil_div(S, 256), mm), dim3(256), 0, st, mk, S, ptr<int>(ctx->srcx));
        LAUNCHCHK();
        if (int e = ensure(ctx, ctx->momout, (size_t)phys_groups(ctx, groups) * ctx->nmom_pad * 2 * ctx->Bpad * 8)) return e;
        ctx->mom_out_arg = ptr<double>(ctx->momout);
        int e = run_xprod(ctx, ptr<int>(ctx->srcx), nullptr, mm, st);
        ctx->mom_out_arg = nullptr;
        if (e) return e;
        // train decompositions
        if (int e2 = run_gram(ctx, mm, false, st)) return e2;
        if (int e2 = ensure(ctx, ctx->Vs, (size_t)mm * Tp * L * 8)) return e2;
        if (int e2 = ensure(ctx, ctx->ds, (size_t)mm * L * 8)) return e2;
        SmallArgs a = small_args(ctx, SMALL_DECOMP);
        a.out_V = ptr<double>(ctx->Vs); a.out_d = ptr<double>(ctx->ds);
        if (int e2 = run_small(ctx, a, mm, st, ptr<double>(ctx->R))) return e2;
        // rescaled copies and offsets, then Q = Rs . Xc^T
        if (int e2 = ensure(ctx, ctx->R2, (size_t)mm * J * ctx->strideR * 8)) return e2;
        if (int e2 = ensure(ctx, ctx->cvc, (size_t)mm * J * Tp * 8)) return e2;
        hipLaunchKernelGGL(k_cv_rescale, dim3(Tp, mm * J), dim3(256), 0, st, ptr<double>(ctx->R), ctx->strideR,
                           ctx->Bpad, ctx->B, J, ctx->npg, ctx->nmom_pad, ptr<double>(ctx->momout),
                           ptr<double>(ctx->R2), ptr<double>(ctx->cvc), Tp, ctx->gps, ptr<int>(ctx->cell_momrow));
        LAUNCHCHK();
        if (int e2 = ensure(ctx, ctx->Qm, (size_t)mm * J * Tp * S * 8)) return e2;
        // Q = Rs . Xc^T (T' x S per split and cell): the product that dominates a split (2 T' S B flop = 1e10 at c4,
        // twice a bootstrap's cross-product).  Round 4: on the 64 x 64-block Gram kernel (P = R . E^T with E = Xc
        // shared by every split: L2 holds it) instead of the generic LDS-tiled NT GEMM
        if (ctx->opt[OPT_GRAM_NT]) {
            if (int e2 = run_nt(ctx, ptr<double>(ctx->R2), ctx->strideR, ctx->Bpad, Tp, ptr<double>(ctx->Xc), 0,
                                ctx->Bpad, S, nullptr, 0, 0, 0, ctx->B, mm * J, ptr<double>(ctx->Qm),
                                (long long)Tp * S, S, nullptr, 0, 0, st))
                return e2;
        } else if (int e2 = run_gram_ex(ctx, mm * J, 2, ptr<double>(ctx->Xc), S, ptr<double>(ctx->Qm), st,
                                        ptr<double>(ctx->R2)))
            return e2;
        if (int e2 = ensure(ctx, ctx->ybar, (size_t)mm * J * T * 8)) return e2;
        if (int e2 = ensure(ctx, ctx->pred, (size_t)mm * S * T * 8)) return e2;
        hipLaunchKernelGGL(k_cv_final, dim3(mm), dim3(256), 0, st, ptr<double>(ctx->Qm), ptr<double>(ctx->cvc),
                           ptr<double>(ctx->Vs), ptr<double>(ctx->ds), ptr<double>(ctx->Y), mk,
                           ptr<int>(ctx->cell_of_row), S, T, J, Tp, L, ptr<double>(ctx->ybar),
                           ptr<double>(ctx->pred), d_r + (size_t)off * T, d_r2 + (size_t)off * T);
        LAUNCHCHK();
    }
    return PLSX_OK;
}
}  // namespace

extern "C" {

// ---- SIMPLS regression (pyls/types/regression.py) ---------------------------
namespace {
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
        hipLaunchKernelGGL(k_sd_post0, grid, block, 0, st, a);
        LAUNCHCHK();
    }
    const size_t lds_step = (size_t)wpb * step_wave;
    HIPCHK(set_lds(k_sd_step, lds_step));
    for (int c = 0; c < k; ++c) {
        a.c = c;
        {
            KTimer tm(ctx, KC_SIMPLS, st);
            hipLaunchKernelGGL(k_sd_step, grid, block, lds_step, st, a);
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
        const size_t lds_f = (size_t)wpb * (k + (a.Vd ? S : 0)) * 8;
        HIPCHK(set_lds(k_sd_final, lds_f));
        hipLaunchKernelGGL(k_sd_final, grid, block, lds_f, st, a);
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
}  // namespace

namespace {
// single-pass form of the SIMPLS bootstrap: signs aligned in dual space (k_sd_final), the feature
// pass accumulates the aligned weights and their squares (k_xprod EPI = 2)
bool simpls_single_pass(const plsx_ctx* ctx)
{
    return (size_t)2 * ctx->ncomp * PLSX_ACC_PITCH * 8 <= 72 * 1024 && ctx->Qs.p && !ctx->opt[OPT_TWO_PASS_BOOT];
}

bool quad_applicable(const plsx_ctx* ctx)
{
    if (!ctx->has_orig) return false;
    return ctx->method == PLSX_REGRESSION ? simpls_single_pass(ctx) : plsc_single_pass(ctx);
}
}  // namespace

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
    // the dual solver runs on batches of up to 4096 bootstraps (whole groups), each followed by the
    // cross-product batches that turn its dual weights into feature-space weights
    int nbs = std::max(nb, (4096 / ctx->npg) * ctx->npg);
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

int plsx_boot_begin(plsx_ctx* ctx, long long n_total, void* stream)
try {
    NEED_ORIG();
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    ctx->quad_active = 0; ctx->quad_n = 0; ctx->series_total = n_total;
    if (n_total < 1 || !quad_applicable(ctx)) return PLSX_OK;
    const int S = ctx->S, L = ctx->method == PLSX_REGRESSION ? ctx->ncomp : ctx->L;
    const int force = ctx->opt[OPT_QUAD_SUMS];                     // 1: whenever applicable, -1: never
    if (force < 0) return PLSX_OK;
    // per-bootstrap pass: 2 S L B n flop; closing pass 2 S^2 L B (rows of C_l in blocks of 384) + 2 S^2 L n for C
    // on the slower tiled GEMM + its transposes: worth it from n ~ 1.25 x the rows the closing pass multiplies
    // (the closing pass contracts a row block of C_l from its own first row on: ~(1 + 1/blocks) / 2 of S^2)
    const int tiles_q = ceil_div(S, 16), gpl_q = quad_blocks(tiles_q);
    const double rows_closing = (double)ceil_div(tiles_q, gpl_q) * gpl_q * 16.0 * 0.5 * (1.0 + 1.0 / gpl_q);
    // ... and S^2 L n for the C_l on the tiled GEMM (symmetric half, at about half the matrix rate): S n / B in the same
    // units (rows of a pass over the B features) -- with few features the per-bootstrap pass is the cheaper one
    const double per_boot = 1.0 - 1.25 * (double)S / std::max(ctx->B, 1);
    if (force == 0 && (per_boot <= 0.0 || (double)n_total * per_boot < 1.25 * rows_closing + 64.0)) return PLSX_OK;
    const size_t cbytes = (size_t)L * S * S * 8;
    // (C_l itself and the partial tiles of the batched S x S products, 2 x 64 x 64 doubles per tile and LV)
    const size_t pbytes = (size_t)L * round_up(S, 64) * round_up(S, 64) * 16;
    if (cbytes + pbytes > (size_t)(0.25 * ctx->scratch_gb * 1073741824.0)) return PLSX_OK;
    if (int e = ensure(ctx, ctx->Cq, cbytes)) return e;
    if (int e = ensure(ctx, ctx->Vsumq, (size_t)L * S * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->Cq.p, 0, cbytes, st));
    HIPCHK(hipMemsetAsync(ctx->Vsumq.p, 0, (size_t)L * S * 8, st));
    ctx->quad_active = 1;
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_boot_finish(plsx_ctx* ctx, double* d_usum, double* d_usq, void* stream)
try {
    NEED_ORIG();
    if (!d_usum || !d_usq) return fail(ctx, PLSX_ERR_ARG, "plsx_boot_finish: null output");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    const int was = ctx->quad_active;
    ctx->quad_active = 0;
    ctx->series_total = 0;
    if (!was || ctx->quad_n == 0) return PLSX_OK;
    ctx->quad_n = 0;
    return quad_finish(ctx, d_usum, d_usq, st);
} PLSX_CATCH(ctx)

int plsx_boot_route(const plsx_ctx* ctx) { return ctx ? ctx->quad_active : 0; }

int plsx_boot_rel(plsx_ctx* ctx, const double* d_orig, const double* d_usum, const double* d_usq,
                  int n_boot, int add_orig, long long count, double* d_bsr, double* d_se, void* stream)
try {
    if (!ctx) return PLSX_ERR_ARG;
    if (!d_orig || !d_usum || !d_usq || !d_bsr || !d_se || count < 1 || n_boot < 1)
        return fail(ctx, PLSX_ERR_ARG, "plsx_boot_rel: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_boot_rel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_orig, d_usum, d_usq, (double)n_boot, add_orig, count,
                       d_bsr, d_se);
    LAUNCHCHK();
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_svd_flip(plsx_ctx* ctx, double* d_xw, double* d_yw, void* stream)
try {
    NEED_DATA();
    if (!d_xw || !d_yw) return fail(ctx, PLSX_ERR_ARG, "plsx_svd_flip: null input");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    const int L = ctx->L;
    // compute.svd decomposes crosscov^T when T' <= B: the flipped factor is then the (B x L) x_weights
    const bool lead_x = ctx->Tp <= ctx->B;
    const double* lead = lead_x ? d_xw : d_yw;
    const long long rows = lead_x ? ctx->B : ctx->Tp;
    if (int e = ensure(ctx, ctx->flipws, (size_t)3 * L * 8)) return e;
    unsigned long long* gmax = ptr<unsigned long long>(ctx->flipws);
    unsigned long long* grow = gmax + L;
    double* signs = reinterpret_cast<double*>(grow + L);
    HIPCHK(hipMemsetAsync(gmax, 0, (size_t)L * 8, st));
    HIPCHK(hipMemsetAsync(grow, 0xff, (size_t)L * 8, st));
    const int nblk = (int)((rows + 4095) / 4096);
    hipLaunchKernelGGL(k_absmax_cols, dim3(nblk), dim3(256), (size_t)L * 8, st, lead, rows, L, gmax);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_argmax_rows, dim3(nblk), dim3(256), 0, st, lead, rows, L, gmax, grow);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_flip_signs, dim3(ceil_div(L, 64)), dim3(64), 0, st, lead, L, grow, signs);
    LAUNCHCHK();
    const long long cx = (long long)ctx->B * L, cy = (long long)ctx->Tp * L;
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((cx + 255) / 256)), dim3(256), 0, st, d_xw, cx, L, signs, d_xw);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((cy + 255) / 256)), dim3(256), 0, st, d_yw, cy, L, signs, d_yw);
    LAUNCHCHK();
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_scale_columns(plsx_ctx* ctx, const double* d_in, long long rows, int cols, const double* d_scale,
                       double* d_out, void* stream)
try {
    if (!ctx) return PLSX_ERR_ARG;
    if (!d_in || !d_scale || !d_out || rows < 1 || cols < 1)
        return fail(ctx, PLSX_ERR_ARG, "plsx_scale_columns: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const long long count = rows * cols;
    hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_in, count, cols, d_scale, d_out);
    LAUNCHCHK();
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_transpose(plsx_ctx* ctx, const double* d_src, int rows, int cols, double* d_dst, void* stream)
try {
    if (!ctx) return PLSX_ERR_ARG;
    if (!d_src || !d_dst || rows < 1 || cols < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_transpose: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_transpose, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(32, 8), 0,
                       static_cast<hipStream_t>(stream), d_src, rows, cols, cols, d_dst, rows);
    LAUNCHCHK();
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_mean_splits(plsx_ctx* ctx, const double* d_in, int np, int ns, int L, double* d_out, void* stream)
try {
    if (!ctx) return PLSX_ERR_ARG;
    if (!d_in || !d_out || np < 1 || ns < 1 || L < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_mean_splits: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_mean_axis1, dim3(ceil_div(np * L, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_in, np, ns, L, d_out);
    LAUNCHCHK();
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_mfma_f64_peak(plsx_ctx* ctx, double* tflops)
try {
    if (!ctx || !tflops) return PLSX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    const int blocks = 256 * 8, iters = 1 << 16;   // ~0.1 s: long enough for the clock to settle
    Buf tmp;
    if (int e = ensure(ctx, tmp, (size_t)blocks * 256 * 8)) return e;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, 0, ptr<double>(tmp), 64);   // warm-up
    HIPCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, 0, ptr<double>(tmp), iters);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    release(tmp);
    const double flops = (double)blocks * 4.0 * iters * 8.0 * 2048.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_percentile_ci(plsx_ctx* ctx, const double* d_data, long long nseries, int n, int i_lo, double g_lo,
                       int i_hi, double g_hi, double* d_lo, double* d_hi, void* stream)
try {
    if (!ctx) return PLSX_ERR_ARG;
    if (!d_data || !d_lo || !d_hi || nseries < 1 || n < 1 || i_lo < 0 || i_hi < 0 || i_lo >= n || i_hi >= n)
        return fail(ctx, PLSX_ERR_ARG, "plsx_percentile_ci: bad arguments");
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    if (p2 > 16384) return fail(ctx, PLSX_ERR_UNSUPPORTED, "plsx_percentile_ci: more than 16384 values per series");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t lds = (size_t)p2 * 8;
    HIPCHK(set_lds(k_percentile2, lds));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int* only = nullptr;
    // long series with both ranks in the tails (the 95 % interval of 10 000 bootstraps): selection instead of a
    // full sort; the few series it cannot settle (pathological pivots) fall through to the sort below
    if (n >= 4096 && std::min(i_lo + 1, n - 1) + 1 <= PSEL_CAP / 2 && n - i_hi <= PSEL_CAP / 2 &&
        !ctx->opt[OPT_PERCENTILE_SORT]) {
        if (int e = ensure(ctx, ctx->pflags, (size_t)nseries * sizeof(int))) return e;
        hipLaunchKernelGGL(k_percentile_sel, dim3((unsigned)nseries), dim3(256), 0, st, d_data, n, i_lo, g_lo, i_hi,
                           g_hi, d_lo, d_hi, ptr<int>(ctx->pflags));
        LAUNCHCHK();
        only = ptr<int>(ctx->pflags);
    }
    hipLaunchKernelGGL(k_percentile2, dim3((unsigned)nseries), dim3(256), lds, st,
                       d_data, n, p2, i_lo, g_lo, i_hi, g_hi, d_lo, d_hi, only);
    LAUNCHCHK();
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_set_scratch(plsx_ctx* ctx, double max_gb, int fixed)
try {
    if (!ctx) return PLSX_ERR_ARG;
    if (!(max_gb > 0.0)) return fail(ctx, PLSX_ERR_ARG, "plsx_set_scratch: budget must be positive");
    if (ctx->has_data) return fail(ctx, PLSX_ERR_STATE, "plsx_set_scratch must precede plsx_set_data");
    ctx->scratch_gb = max_gb;
    ctx->scratch_fixed = fixed ? 1 : 0;
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_set_timing(plsx_ctx* ctx, int enable)
try {
    if (!ctx) return PLSX_ERR_ARG;
    ctx->timing = enable ? 1 : 0;
    for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); }
    ctx->events.clear();
    ctx->timed_units = 0;
    ctx->nt_flops = 0.0;
    ctx->quad_series = 0;
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_last_timing(const plsx_ctx* cctx, double* out, int cap)
try {
    plsx_ctx* ctx = const_cast<plsx_ctx*>(cctx);
    if (!ctx || !out || cap < 1) return PLSX_ERR_ARG;
    double ms = 0.0;
    int launches = 0;
    for (auto& ev : ctx->events) {
        if (ev.cls != KC_XPROD) continue;
        float t = 0.f;
        if (hipEventSynchronize(ev.e1) == hipSuccess && hipEventElapsedTime(&t, ev.e0, ev.e1) == hipSuccess)
            ms += t;
        ++launches;
    }
    // compact blocks: k-steps x 4 rows a block of the last launch contracted over, as a fraction of S (0: the
    // last launch was not compact) -- the issued share of the dense S-row contraction
    double crows = 0.0;
    if (ctx->last_compact_n > 0) {
        std::vector<int> cnt(ctx->last_compact_n);
        if (hipMemcpy(cnt.data(), ptr<int>(ctx->rowtab_c) + (size_t)ctx->last_compact_n * ctx->last_compact_ktot,
                      cnt.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int c : cnt) crows += 4.0 * ((std::max(c, 1) + 3) / 4);
            crows /= (double)cnt.size() * ctx->S;
        }
    }
    const bool cmp = ctx->last_compact_n > 0;
    double vals[12] = {ms, (double)launches, (double)(cmp ? 1 : (ctx->sepmom_used ? ctx->npg_d : ctx->npg)),
                       (double)(cmp ? ceil_div(ctx->Tp, 16) : (ctx->sepmom_used ? ctx->MTd : ctx->MT)),
                       (double)ctx->Gcap * ctx->npg, (double)ctx->timed_units, (double)use_dual(ctx), crows,
                       ctx->nt_flops, (double)ctx->quad_series, (double)ctx->quad_MT, (double)ctx->quad_gpl};
    int n = std::min(cap, 12);
    for (int i = 0; i < n; ++i) out[i] = vals[i];
    return n;
} PLSX_CATCH(const_cast<plsx_ctx*>(cctx))

int plsx_kernel_timing(const plsx_ctx* cctx, int kernel_class, double* ms_out, int* launches_out)
try {
    plsx_ctx* ctx = const_cast<plsx_ctx*>(cctx);
    if (!ctx || kernel_class < 0 || kernel_class >= KC_COUNT) return PLSX_ERR_ARG;
    double ms = 0.0;
    int launches = 0;
    for (auto& ev : ctx->events) {
        if (ev.cls != kernel_class) continue;
        float t = 0.f;
        if (hipEventSynchronize(ev.e1) == hipSuccess && hipEventElapsedTime(&t, ev.e0, ev.e1) == hipSuccess)
            ms += t;
        ++launches;
    }
    if (ms_out) *ms_out = ms;
    if (launches_out) *launches_out = launches;
    return PLSX_OK;
} PLSX_CATCH(const_cast<plsx_ctx*>(cctx))

const char* plsx_kernel_class_name(int kernel_class)
{
    return (kernel_class >= 0 && kernel_class < KC_COUNT) ? kKernelClassNames[kernel_class] : nullptr;
}

int plsx_set_perm_path(plsx_ctx* ctx, int dual)
try {
    NEED_DATA();
    if (dual >= 0) ctx->dual = (dual && ctx->dual_ok) ? 1 : 0;      // dual < 0: keep the route
    ctx->has_Kd = 0;            // the next dual call forms K again (bench.py: once per timed analysis)
    return use_dual(ctx);
} PLSX_CATCH(ctx)

int plsx_set_option(plsx_ctx* ctx, const char* key, int value)
try {
    if (!ctx || !key) return PLSX_ERR_ARG;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(key, kOptionNames[i])) {
            // layout-time switches are read by plsx_set_data
            const bool plan_time = (i == OPT_XPROD_MT24 || i == OPT_MIN_BATCH || i == OPT_INBLOCK_MOMENTS ||
                                    i == OPT_NO_FIXED_X || i == OPT_NO_DUAL_PERM);
            if (plan_time && ctx->has_data && ctx->opt[i] != value)
                return fail(ctx, PLSX_ERR_STATE, std::string("plsx_set_option: '") + key + "' must precede plsx_set_data");
            ctx->opt[i] = value;
            return PLSX_OK;
        }
    return fail(ctx, PLSX_ERR_ARG, std::string("plsx_set_option: unknown option '") + key + "'");
} PLSX_CATCH(ctx)

const char* plsx_option_name(int index)
{
    return (index >= 0 && index < OPT_COUNT) ? kOptionNames[index] : nullptr;
}

int plsx_numeric_report(plsx_ctx* ctx, long long* refined, long long* unrefined)
try {
    if (!ctx) return PLSX_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    int stw[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(stw, ctx->status.p, 4 * sizeof(int), hipMemcpyDeviceToHost));
    if (stw[1] || stw[2]) HIPCHK(hipMemset(static_cast<int*>(ctx->status.p) + 1, 0, 2 * sizeof(int)));
    if (refined) *refined = ctx->n_refined + stw[1];
    if (unrefined) *unrefined = ctx->n_unrefined + stw[2];
    ctx->n_refined = ctx->n_unrefined = 0;
    return PLSX_OK;
} PLSX_CATCH(ctx)

// ---- host-side index generators (no device, no context) ------------------------
namespace {
int load_mt(plsx_rs::MT& rs, const uint32_t* key, int pos)
{
    if (!key || pos < 0 || pos > 624) return PLSX_ERR_ARG;
    memcpy(rs.key, key, sizeof(rs.key));
    rs.pos = pos;
    return 0;
}
bool bad_design(const int* groups, int n_groups, int n_cond)
{
    if (!groups || n_groups < 1 || n_cond < 1) return true;
    for (int i = 0; i < n_groups; ++i) if (groups[i] < 1) return true;
    return false;
}
}  // namespace

int plsx_gen_permsamp_stream(const int* groups, int n_groups, int n_cond, int n_perm, uint32_t* mt_key, int* mt_pos,
                             int32_t* out, int* rows_done)
try {
    plsx_rs::MT rs;
    if (bad_design(groups, n_groups, n_cond) || n_perm < 0 || !out || !mt_pos || load_mt(rs, mt_key, *mt_pos))
        return PLSX_ERR_ARG;
    const int w = plsx_rs::gen_permsamp(plsx_rs::Design(groups, n_groups, n_cond), n_perm, rs, out, rows_done);
    memcpy(mt_key, rs.key, sizeof(rs.key));
    *mt_pos = rs.pos;
    return w;
} PLSX_CATCH(nullptr)

int plsx_gen_permsamp(const int* groups, int n_groups, int n_cond, int n_perm, uint32_t* mt_key, int* mt_pos,
                      int32_t* out)
try {
    return plsx_gen_permsamp_stream(groups, n_groups, n_cond, n_perm, mt_key, mt_pos, out, nullptr);
} PLSX_CATCH(nullptr)

int plsx_gen_bootsamp_stream(const int* groups, int n_groups, int n_cond, int n_boot, uint32_t* mt_key, int* mt_pos,
                             int32_t* out, int* rows_done)
try {
    plsx_rs::MT rs;
    if (bad_design(groups, n_groups, n_cond) || n_boot < 0 || !out || !mt_pos || load_mt(rs, mt_key, *mt_pos))
        return PLSX_ERR_ARG;
    const int w = plsx_rs::gen_bootsamp(plsx_rs::Design(groups, n_groups, n_cond), n_boot, rs, out, rows_done);
    memcpy(mt_key, rs.key, sizeof(rs.key));
    *mt_pos = rs.pos;
    return w;
} PLSX_CATCH(nullptr)

int plsx_gen_bootsamp(const int* groups, int n_groups, int n_cond, int n_boot, uint32_t* mt_key, int* mt_pos,
                      int32_t* out)
try {
    return plsx_gen_bootsamp_stream(groups, n_groups, n_cond, n_boot, mt_key, mt_pos, out, nullptr);
} PLSX_CATCH(nullptr)

int plsx_gen_splits(const int* groups, int n_groups, int n_cond, int n_split, double test_size, uint32_t* mt_key,
                    int* mt_pos, uint8_t* out)
try {
    plsx_rs::MT rs;
    if (bad_design(groups, n_groups, n_cond) || n_split < 0 || !out || !mt_pos || load_mt(rs, mt_key, *mt_pos))
        return PLSX_ERR_ARG;
    const int w = plsx_rs::gen_splits(plsx_rs::Design(groups, n_groups, n_cond), n_split, test_size, rs, out);
    memcpy(mt_key, rs.key, sizeof(rs.key));
    *mt_pos = rs.pos;
    return w;
} PLSX_CATCH(nullptr)

int plsx_gen_splits_seeded(const int* groups, int n_groups, int n_cond, int n_split, double test_size,
                           const uint32_t* seeds, int n_seeds, uint8_t* out)
try {
    if (bad_design(groups, n_groups, n_cond) || n_split < 0 || n_seeds < 0 || !out || (n_seeds && !seeds))
        return PLSX_ERR_ARG;
    const plsx_rs::Design d(groups, n_groups, n_cond);
    // independent streams: spread over host threads
    const int nth = (int)std::max(1u, std::min({std::thread::hardware_concurrency(), 32u, (unsigned)((n_seeds + 15) / 16)}));
    std::vector<int> warn(nth, 0);
    auto work = [&](int t) {
        for (int i = t; i < n_seeds; i += nth) {
            plsx_rs::MT rs;
            rs.seed(seeds[i]);
            warn[t] |= plsx_rs::gen_splits(d, n_split, test_size, rs, out + (size_t)i * n_split * d.n_rows);
        }
    };
    // thread creation may throw (ulimit, container limits) and nothing may unwind through the
    // C ABI: the streams of a worker that could not be started are drawn on this thread
    std::vector<std::thread> pool;
    std::vector<int> inline_work;
    for (int t = 1; t < nth; ++t) {
        try {
            pool.emplace_back(work, t);
        } catch (...) {
            inline_work.push_back(t);
        }
    }
    work(0);
    for (int t : inline_work) work(t);
    for (auto& th : pool) th.join();
    int w = 0;
    for (int v : warn) w |= v;
    return w;
} PLSX_CATCH(nullptr)

}  // extern "C"
