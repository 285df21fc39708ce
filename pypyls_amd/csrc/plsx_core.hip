// plsx_core.hip -- context, planning, plsx_set_data, permutations, PLS-C bootstraps, finishing, index generators
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_internal.h"
#include "plsx_resample.h"
using namespace plsxi;

namespace plsxi {

const char* const kKernelClassNames[KC_COUNT] = {"k_xprod", "k_gram", "k_small", "k_urot", "k_nt_gemm",
                                                 "k_ucorr_partial", "k_simpls_dual", "k_build_A", "k_xprod_moments"};

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
    if (c->method == PLSX_BEHAVIORAL) {
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
    // what mapping a GB costs HERE was measured when the data were bound (probe_map_cost, plsx_set_data), in classes
    if (ctx->map_ms_per_gb <= 0.0) ctx->map_ms_per_gb = 40.0;
    const double c_group = ctx->map_ms_per_gb * gb_per_group, c_launch = ctx->Tp > PLSX_JACOBI_TP ? std::max(2.5, 25.0 * tn * tn * tn) : 2.5;
    int g = round_up((int)std::ceil(std::sqrt(c_launch * (double)need / std::max(c_group, 1e-3))), 8);
    g = std::max(g, ctx->Galloc);
    // a re-bound context (the front-ends' cached engine) keeps the R it mapped for an earlier call: groups that fit
    // what is already there cost nothing to map
    if (ctx->R.bytes > 0 && gb_per_group > 0.0)
        g = std::max(g, (int)std::min<double>(cap, std::floor((double)ctx->R.bytes / (gb_per_group * 1073741824.0))));
    return std::max(1, std::min(g, cap));
}

// What mapping a GB of device memory costs on THIS device right now: ~1 ms per GB on clean pages (the model of
// launch_groups then launches 2 - 3 x larger super-batches), 25+ ms per GB when the driver has to clear recycled VRAM
// first.  Measured once per context at bind time -- outside the launch path: the probe ends in a device-wide sync --
// on 1 GB (a smaller request is served from the pool of clean pages and sees nothing), and QUANTISED to four classes so that the super-batch sizes (and with them the summation order of the
// bootstrap sums) do not follow measurement noise from run to run.  A failed probe (full device) keeps the
// conservative 40 ms per GB.
void probe_map_cost(plsx_ctx* ctx)
{
    if (ctx->map_ms_per_gb > 0.0 || ctx->scratch_fixed) return;
    ctx->map_ms_per_gb = 40.0;
    void* probe = nullptr;
    const size_t pb = (size_t)1 << 30;
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    if (hipMalloc(&probe, pb) == hipSuccess) {
        (void)hipMemset(probe, 0, pb);
        (void)hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        (void)hipFree(probe);
        const double per_gb = 2.0 * ms;                    // (safety factor 2 on the per-GB time)
        ctx->map_ms_per_gb = per_gb < 3.0 ? 1.5 : (per_gb < 10.0 ? 5.0 : (per_gb < 25.0 ? 15.0 : 40.0));
    } else (void)hipGetLastError();
}

// Super-batches of a call of n resamples with at most `cap` per launch (cap a multiple of `per_group`): the same
// number of launches, but of equal size -- 625 bootstraps run as 315 + 310, not 504 + 121 (a launch of 121 costs
// the latency-bound stages, one wave of the small solver and the moment blocks, as much as one of 504).
int balanced_batch(int n, int cap, int per_group)
{
    const int launches = ceil_div(n, std::max(cap, 1));
    return std::min(cap, round_up(ceil_div(n, launches), std::max(per_group, 1)));
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

bool quad_applicable(const plsx_ctx* ctx)
{
    if (!ctx->has_orig) return false;
    return ctx->method == PLSX_REGRESSION ? simpls_single_pass(ctx) : plsc_single_pass(ctx);
}

}  // namespace plsxi

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
    (void)plsx_comm_destroy(ctx);
    for (Buf* b : {&ctx->Xc, &ctx->xmean, &ctx->Y, &ctx->cell_of_row, &ctx->cell_start, &ctx->cell_len,
                   &ctx->out_row, &ctx->mom_idx, &ctx->mom_n, &ctx->Afrag, &ctx->R, &ctx->Gm, &ctx->Pm,
                   &ctx->part, &ctx->Mfrag, &ctx->U0T, &ctx->V0, &ctx->d0, &ctx->tmpW,
                   &ctx->Rfull, &ctx->Vp, &ctx->dp, &ctx->Mvd, &ctx->Cm, &ctx->srcx, &ctx->srcy, &ctx->part2,
                   &ctx->Kmat, &ctx->swork, &ctx->spct, &ctx->sc,
                   &ctx->momout, &ctx->R2, &ctx->cvc, &ctx->Qm, &ctx->Vs, &ctx->ds, &ctx->ybar, &ctx->pred,
                   &ctx->Xn, &ctx->out_row_f, &ctx->mom_idx_f, &ctx->Kd, &ctx->Ad, &ctx->Wd, &ctx->gws, &ctx->cellS, &ctx->rowc, &ctx->out_row_s, &ctx->okx, &ctx->oky, &ctx->psum, &ctx->psq, &ctx->row_slice, &ctx->row_local, &ctx->slice_cell0, &ctx->cell_momrow, &ctx->status, &ctx->ScT, &ctx->out_row_w, &ctx->Qs, &ctx->out_row_d, &ctx->mom_idx_d, &ctx->Afrag_m, &ctx->momn_m, &ctx->scale,
                   &ctx->Afrag_c, &ctx->rank_c, &ctx->rowtab_c, &ctx->m1_c, &ctx->m2_c, &ctx->out_row_c, &ctx->mom_idx_c, &ctx->mask_c,
                   &ctx->refV, &ctx->refLam, &ctx->refK0, &ctx->refPart, &ctx->refPartP, &ctx->refH, &ctx->flipws, &ctx->pflags,
                   &ctx->Cq, &ctx->Vsumq, &ctx->Vdq, &ctx->Vtq, &ctx->Afrag_q, &ctx->qpart, &ctx->ccon, &ctx->sFt, &ctx->Yrot})
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
    if (method == PLSX_REGRESSION && simpls_step_lds_bytes(S, T, ncomp) > 158 * 1024)
        return fail(ctx, PLSX_ERR_UNSUPPORTED,
                    "SIMPLS: S / T too large for the on-chip component step (8 (T^2 + S) bytes must fit 158 KB)");
    // 32-bit buffer offsets (the kernels address one resample's cross-covariance matrix R_r, T'pp x Bpad doubles, through a
    // buffer resource whose byte offsets are 31 bits; offsets beyond read as zero): ONE R_r must stay below 2 GB.  That is
    // the real limit -- it allows 5.1 million features at the headline T' = 50 and 16 million for mean-centred designs,
    // and it refuses e.g. T' = 200 with 1.4 million features, which the old flat "2,000,000 columns" rule let through
    // to kernels that would have read zeros.  (Columns alone: a 4-row k-step of X, 32 Bpad bytes, fits up to 67 million.)
    {
        const long long tpp = ((long long)Tp + 3) / 4 * 4, bpad = ((long long)B + std::min(Tp, B) + 127) / 128 * 128;
        if (tpp * bpad * 8 >= (1LL << 31) || bpad > 16000000LL) {
            char msg[256];
            snprintf(msg, sizeof msg, "T' x B too large: one cross-covariance matrix (%lld x %lld doubles = %.2f GB) must "
                     "stay below 2 GB, and B below 16,000,000 (32-bit buffer offsets)", tpp, bpad, tpp * bpad * 8 / 1073741824.0);
            return fail(ctx, PLSX_ERR_UNSUPPORTED, msg);
        }
    }
    if (Tp > PLSX_MAX_TP || J > PLSX_MAX_CELLS || (method != PLSX_BEHAVIORAL && Tp > PLSX_BLOCK_TP)) {
        char msg[200];
        snprintf(msg, sizeof msg, "stacked dimension T' = %d (cells J = %d) exceeds the limit (T' <= %d for "
                 "behavioral PLS, %d otherwise; J <= %d)", Tp, J, PLSX_MAX_TP, PLSX_BLOCK_TP, PLSX_MAX_CELLS);
        return fail(ctx, PLSX_ERR_UNSUPPORTED, msg);
    }
    ctx->has_data = ctx->has_orig = false;
    // a new binding starts a new analysis: numerical status and graded-spectrum counters of the last one are dropped
    ctx->n_refined = ctx->n_unrefined = 0;
    HIPCHK(hipMemsetAsync(ctx->status.p, 0, 4 * sizeof(int), st));
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
        // (the key holds everything a slot's padding depends on: rows T'..T'pp, columns B + L..Bpad, and the method --
        // a route of another method may have left other rows / columns of a slot untouched)
        const long long geom[6] = {Tp_n, round_up(Tp_n, 4), round_up(B + L_n, 128), B, L_n, method * 2 + ((flags & PLSX_FLAG_COVARIANCE) ? 1 : 0)};
        bool same = ctx->R.p && ctx->R_zeroed_bytes == ctx->R.bytes;
        for (int i = 0; i < 6; ++i) same = same && ctx->R_geom[i] == geom[i];
        if (ctx->R.p && !same) HIPCHK(hipMemsetAsync(ctx->R.p, 0, ctx->R.bytes, st));
        for (int i = 0; i < 6; ++i) ctx->R_geom[i] = geom[i];
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
    probe_map_cost(ctx);
    if (plan_groups(ctx) != 0)
        return fail(ctx, PLSX_ERR_UNSUPPORTED,
                    "cannot lay out the rows of a resample (with their per-cell moment rows) over cross-product blocks");
    if (int e = upload_rowmaps(ctx)) return e;
    if (int e = plan_sepmom(ctx)) return e;
    ctx->fix = 0; ctx->has_Xn = 0; ctx->npgf = 0; ctx->group_stride_f = 0; ctx->has_cellS = 0; ctx->has_sFt = 0;
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
    const bool fix = ctx->Tp > 1 && !ctx->opt[OPT_NO_REFINE];
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

}  // extern "C"

namespace plsxi {

int perm_batch_impl(plsx_ctx* ctx, const int32_t* d_perm_idx, const double* d_ystack, int n, int rotate,
                    double* d_out_sv, void* stream);

}  // namespace plsxi

extern "C" {

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

}  // extern "C"

namespace plsxi {

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
                               ptr<int>(ctx->cell_of_row), idx, lay, ptr<double>(ctx->Ad), (size_t)0, Sd,
                               -1);                           // (a permutation repeats no row: plain stores)
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

}  // namespace plsxi

namespace plsxi {

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

}  // namespace plsxi

namespace plsxi {

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
            int cap = 0;
            if (ctx->method == PLSX_BEHAVIORAL) {
                const size_t lds = build_lds_behav(ctx, &cap);
                hipLaunchKernelGGL(k_build_A_behav, dim3(m, ctx->J), dim3(256), lds, st,
                                   ptr<double>(ctx->Y), 0LL, ctx->T, S, ptr<int>(ctx->cell_start),
                                   ptr<int>(ctx->cell_len), idx, idx, lay, ctx->cov, 0, ptr<double>(ctx->Ad),
                                   (size_t)0, (double*)nullptr, 0, Sd, (double*)nullptr, (size_t)0, (const int*)nullptr,
                                   PLSX_MOM_PAIRS, cap);
            } else {
                const size_t lds = build_lds_mc(ctx, &cap);
                hipLaunchKernelGGL(k_build_A_mc, dim3(m), dim3(256), lds, st, S, ctx->J, ctx->n_cond, ctx->mc,
                                   ptr<int>(ctx->cell_of_row), idx, lay, ptr<double>(ctx->Ad), (size_t)0, Sd, cap);
            }
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

}  // namespace plsxi

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
    // compute.svd decomposes crosscov^T when T' <= B: the flipped factor is then the (B x L) x_weights.  SIMPLS
    // (pyls/types/regression.py:103: compute.svd of Cov = Y0^T X0 (T x B) per component, flipped per compute.py:43-50): on the x side when
    // B > T, else on the right vectors c (T x k); L = n_components there
    const bool reg = ctx->method == PLSX_REGRESSION;
    const int L = reg ? ctx->ncomp : ctx->L;
    const int Ty = reg ? ctx->T : ctx->Tp;
    const bool lead_x = reg ? ctx->B > ctx->T : ctx->Tp <= ctx->B;
    const double* lead = lead_x ? d_xw : d_yw;
    const long long rows = lead_x ? ctx->B : Ty;
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
    const long long cx = (long long)ctx->B * L, cy = (long long)Ty * L;
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

int plsx_center_rows(plsx_ctx* ctx, const double* d_in, int rows, long long cols, double* d_out, void* stream)
try {
    if (!ctx) return PLSX_ERR_ARG;
    if (!d_in || !d_out || rows < 1 || cols < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_center_rows: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_center_rows, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), d_in, cols, d_out);
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
            const bool plan_time = (i == OPT_MIN_BATCH || i == OPT_INBLOCK_MOMENTS ||
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

