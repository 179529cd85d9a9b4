// iamr_amd/csrc/cabi.hip -- extern "C" boundary of libiamrx.so (declarations: include/iamrx.h).
// Exceptions never cross the boundary: they become a non-zero return code + iamrx_last_error().
#include "../../include/iamrx.h"
#include "operators.h"
#include "launch.h"
#include "amrns.h"
#include <string>
#include <cstring>
#include <vector>
#include <map>

using namespace iamrx;

struct iamrx_layout_s { LayoutP p; };
struct iamrx_mf_s { MultiFab mf; };

static thread_local std::string g_err;

#define IAMRX_TRY try {
#define IAMRX_CATCH                                                   \
    return 0; }                                                       \
    catch (const std::exception& e) { g_err = e.what(); return 1; }   \
    catch (...) { g_err = "unknown error"; return 2; }

static Geometry to_geom(const iamrx_geom* g)
{
    Geometry r;
    for (int d = 0; d < 3; ++d) {
        r.domain.lo[d] = g->dom_lo[d]; r.domain.hi[d] = g->dom_hi[d];
        r.problo[d] = g->prob_lo[d]; r.probhi[d] = g->prob_hi[d];
        r.periodic[d] = g->periodic[d];
        r.dx[d] = (g->prob_hi[d] - g->prob_lo[d]) / (double)(g->dom_hi[d] - g->dom_lo[d] + 1);
    }
    return r;
}

static MGOpts to_opts(const iamrx_mg_opts* o)
{
    MGOpts r;
    if (!o) return r;
    r.nu1 = o->nu1; r.nu2 = o->nu2; r.nuf = o->nuf; r.nub = o->nub;
    r.max_iters = o->max_iters; r.bottom_maxiter = o->bottom_maxiter; r.bottom_reltol = o->bottom_reltol;
    r.omega = o->omega; r.maxorder = o->maxorder; r.max_coarsening_level = o->max_coarsening_level;
    r.min_width = o->min_width; r.nodal_sweeps = o->nodal_sweeps; r.nodal_smoother = o->nodal_smoother;
    r.verbose = o->verbose; r.bottom_smoother_only = o->bottom_smoother_only; r.fixed_iters = o->fixed_iters;
    r.nodal_nu1 = o->nodal_nu1; r.nodal_nu2 = o->nodal_nu2; r.device_bottom = o->device_bottom; r.slab = o->slab;
    return r;
}

static void from_stats(const MGStats& s, iamrx_mg_stats* o)
{
    if (!o) return;
    o->iters = s.iters; o->resnorm0 = s.resnorm0; o->rhsnorm0 = s.rhsnorm0; o->resnorm = s.resnorm;
    o->bottom_iters_total = s.bottom_iters_total; o->converged = s.converged; o->vcycle_ms = s.vcycle_ms; o->nlevels = s.nlevels;
}

static DomainBC to_bc(const int lobc[3], const int hibc[3], int maxorder)
{
    DomainBC b;
    for (int d = 0; d < 3; ++d) { b.lo[d] = lobc ? lobc[d] : 0; b.hi[d] = hibc ? hibc[d] : 0; }
    b.maxorder = maxorder;
    return b;
}

extern "C" {

const char* iamrx_last_error(void) { return g_err.c_str(); }

int iamrx_init(int device) { IAMRX_TRY Context::get().init(device); IAMRX_CATCH }
int iamrx_finalize(void) { IAMRX_TRY Context::get().release_cache(); IAMRX_CATCH }
int iamrx_sync(void) { IAMRX_TRY Context::get().sync(); IAMRX_CATCH }
void* iamrx_stream(void) { return (void*)Context::get().stream; }
int iamrx_tuning_set(const char* key, double value) { IAMRX_TRY if (!key) throw Error("iamrx_tuning_set: null key"); tuning_set(key, value); IAMRX_CATCH }
int iamrx_tuning_get(const char* key, double default_value, double* value) { IAMRX_TRY *value = tune_by_name(key, default_value); IAMRX_CATCH }
int iamrx_scope_profile(int enable, int reset, char* report, size_t capacity)
{
    IAMRX_TRY
    if (report && capacity > 0) { const std::string r = scope_profile_report(); strncpy(report, r.c_str(), capacity - 1); report[capacity - 1] = 0; }
    if (enable >= 0) scope_profile_enable(enable != 0, reset != 0);
    IAMRX_CATCH
}
int iamrx_coalesce_merge_count(size_t* n) { IAMRX_TRY *n = coalesce_merge_count(); IAMRX_CATCH }
int iamrx_exchange_counts(size_t out[4])
{
    IAMRX_TRY
    auto& c = Context::get();
    out[0] = c.n_exchange[0]; out[1] = c.exchange_doubles[0]; out[2] = c.n_exchange[1]; out[3] = c.exchange_doubles[1];
    IAMRX_CATCH
}
int iamrx_sync_count(size_t* n_stream_sync) { IAMRX_TRY *n_stream_sync = Context::get().n_stream_sync; IAMRX_CATCH }
int iamrx_alloc_count(size_t* n_device_malloc)
{
    IAMRX_TRY
    *n_device_malloc = Context::get().n_device_malloc;
    IAMRX_CATCH
}
int iamrx_mem_info(size_t* live, size_t* cached)
{
    IAMRX_TRY
    if (live) *live = Context::get().bytes_live;
    if (cached) *cached = Context::get().bytes_cached;
    IAMRX_CATCH
}

// host-only (no device needed): the ghost-exchange plan rank `rank` would execute for a level.
// desc layout (16 ints): kind (0 local copy, 1 pack+send, 2 recv+unpack), peer rank (-1 local), src global box (-1 recv),
// dst global box (-1 send), region lo[3], hi[3] (destination index frame), shift[3] (src = dst + shift), buf_off (points), pad
int iamrx_host_fill_plan(int nboxes, const int* lo_hi, const int* owner, int rank, const int type[3], int ngrow, const iamrx_geom* g,
                         int max_desc, int* desc, int* ndesc)
{
    return iamrx_host_fill_plan_wall_ext(nboxes, lo_hi, owner, rank, type, ngrow, g, 0, max_desc, desc, ndesc);
}
int iamrx_host_fill_plan_wall_ext(int nboxes, const int* lo_hi, const int* owner, int rank, const int type[3], int ngrow, const iamrx_geom* g,
                                  int wall_ext, int max_desc, int* desc, int* ndesc)
{
    IAMRX_TRY
    std::vector<BoxD> b(nboxes);
    std::vector<int> own(nboxes), local_of(nboxes, -1), local;
    for (int i = 0; i < nboxes; ++i) {
        for (int d = 0; d < 3; ++d) { b[i].lo[d] = lo_hi[6 * i + d]; b[i].hi[d] = lo_hi[6 * i + 3 + d]; }
        own[i] = owner ? owner[i] : 0;
        if (own[i] == rank) { local_of[i] = (int)local.size(); local.push_back(i); }
    }
    CopyPlan plan;
    std::map<int, CopyPlan::Peer> peers;
    IndexType t{{type[0], type[1], type[2]}};
    build_fill_plan_host(b, own, local_of, rank, t, ngrow, to_geom(g), plan, peers, nullptr, -1, wall_ext);
    int n = 0;
    auto emit = [&](int kind, int peer, const CopyDesc& cd) {
        if (n < max_desc && desc) {
            int* o = desc + 16 * n;
            o[0] = kind; o[1] = peer;
            o[2] = cd.src_fab >= 0 ? local[cd.src_fab] : -1;
            o[3] = cd.dst_fab >= 0 ? local[cd.dst_fab] : -1;
            for (int d = 0; d < 3; ++d) { o[4 + d] = cd.region.lo[d]; o[7 + d] = cd.region.hi[d]; o[10 + d] = cd.shift[d]; }
            o[13] = (int)cd.buf_off; o[14] = 0; o[15] = 0;
        }
        ++n;
    };
    for (auto& cd : plan.local) emit(0, -1, cd);
    for (auto& kv : peers) {
        for (auto& cd : kv.second.pack) emit(1, kv.first, cd);
        for (auto& cd : kv.second.unpack) emit(2, kv.first, cd);
    }
    *ndesc = n;
    IAMRX_CATCH
}

static hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
int iamrx_timer_start(void)
{
    IAMRX_TRY
    if (!g_ev0) { IAMRX_HIP_CHECK(hipEventCreate(&g_ev0)); IAMRX_HIP_CHECK(hipEventCreate(&g_ev1)); }
    IAMRX_HIP_CHECK(hipEventRecord(g_ev0, Context::get().stream));
    IAMRX_CATCH
}
int iamrx_timer_stop(double* ms)
{
    IAMRX_TRY
    IAMRX_HIP_CHECK(hipEventRecord(g_ev1, Context::get().stream));
    IAMRX_HIP_CHECK(hipEventSynchronize(g_ev1));
    float f = 0.f;
    IAMRX_HIP_CHECK(hipEventElapsedTime(&f, g_ev0, g_ev1));
    *ms = (double)f;
    IAMRX_CATCH
}

void iamrx_mg_default_opts(iamrx_mg_opts* o)
{
    MGOpts d;
    o->nu1 = d.nu1; o->nu2 = d.nu2; o->nuf = d.nuf; o->nub = d.nub; o->max_iters = d.max_iters;
    o->bottom_maxiter = d.bottom_maxiter; o->bottom_reltol = d.bottom_reltol; o->omega = d.omega; o->maxorder = d.maxorder;
    o->max_coarsening_level = d.max_coarsening_level; o->min_width = d.min_width; o->nodal_sweeps = d.nodal_sweeps;
    o->nodal_smoother = d.nodal_smoother; o->verbose = d.verbose; o->bottom_smoother_only = d.bottom_smoother_only;
    o->fixed_iters = d.fixed_iters; o->nodal_nu1 = d.nodal_nu1; o->nodal_nu2 = d.nodal_nu2; o->device_bottom = d.device_bottom; o->slab = d.slab;
}

int iamrx_layout_create(int nboxes, const int* lo_hi, const int* owner, iamrx_layout* out)
{
    IAMRX_TRY
    IAMRX_ASSERT(Context::get().stream != nullptr);
    std::vector<BoxD> b(nboxes);
    std::vector<int> own(nboxes);
    for (int i = 0; i < nboxes; ++i) {
        for (int d = 0; d < 3; ++d) { b[i].lo[d] = lo_hi[6 * i + d]; b[i].hi[d] = lo_hi[6 * i + 3 + d]; }
        own[i] = owner ? owner[i] : 0;
    }
    auto* h = new iamrx_layout_s;
    h->p = std::make_shared<Layout>(b, own, Context::get().comm->rank);
    *out = h;
    IAMRX_CATCH
}
// the layout the level objects work on for the caller's boxes: boxes of one owner that share full faces merged (mf.h: coalesce_layout);
// nboxes / lo_hi as in iamrx_amr_level_boxes (lo_hi NULL: query the count)
int iamrx_layout_coalesced_boxes(iamrx_layout l, int* nboxes, int* lo_hi)
{
    IAMRX_TRY
    LayoutP c = coalesce_layout(l->p);
    const auto& bx = c->boxes;
    if (lo_hi) {
        if (*nboxes < (int)bx.size()) throw Error("iamrx_layout_coalesced_boxes: box capacity too small");
        for (size_t q = 0; q < bx.size(); ++q) for (int d = 0; d < 3; ++d) { lo_hi[6 * q + d] = bx[q].lo[d]; lo_hi[6 * q + 3 + d] = bx[q].hi[d]; }
    }
    *nboxes = (int)bx.size();
    IAMRX_CATCH
}
int iamrx_layout_destroy(iamrx_layout l) { IAMRX_TRY delete l; IAMRX_CATCH }
int iamrx_layout_nlocal(iamrx_layout l, int* n) { IAMRX_TRY *n = l->p->nlocal(); IAMRX_CATCH }
int iamrx_layout_local_box(iamrx_layout l, int li, int lo_hi[6], int* gi)
{
    IAMRX_TRY
    const BoxD& b = l->p->lbox(li);
    for (int d = 0; d < 3; ++d) { lo_hi[d] = b.lo[d]; lo_hi[3 + d] = b.hi[d]; }
    if (gi) *gi = l->p->local[li];
    IAMRX_CATCH
}

int iamrx_mf_create(iamrx_layout l, const int type[3], int ncomp, int ngrow, iamrx_mf* out)
{
    IAMRX_TRY
    IndexType t{{type[0], type[1], type[2]}};
    auto* h = new iamrx_mf_s;
    h->mf.define(l->p, t, ncomp, ngrow);
    *out = h;
    IAMRX_CATCH
}
int iamrx_mf_destroy(iamrx_mf m) { IAMRX_TRY delete m; IAMRX_CATCH }
int iamrx_mf_info(iamrx_mf m, int* ncomp, int* ngrow, int type[3], int* nlocal)
{
    IAMRX_TRY
    if (ncomp) *ncomp = m->mf.ncomp;
    if (ngrow) *ngrow = m->mf.ngrow;
    if (type) for (int d = 0; d < 3; ++d) type[d] = m->mf.type.t[d];
    if (nlocal) *nlocal = m->mf.nlocal();
    IAMRX_CATCH
}
int iamrx_mf_fab_box(iamrx_mf m, int li, int lo_hi[6])
{
    IAMRX_TRY
    BoxD b = m->mf.fabbox(li);
    for (int d = 0; d < 3; ++d) { lo_hi[d] = b.lo[d]; lo_hi[3 + d] = b.hi[d]; }
    IAMRX_CATCH
}
int iamrx_mf_dev_ptr(iamrx_mf m, int li, double** p) { IAMRX_TRY *p = m->mf.h_tab.at(li).p; IAMRX_CATCH }
int iamrx_mf_to_host(iamrx_mf m, int li, double* dst) { IAMRX_TRY m->mf.copy_to_host(li, dst); IAMRX_CATCH }
int iamrx_mf_from_host(iamrx_mf m, int li, const double* src) { IAMRX_TRY m->mf.copy_from_host(li, src); IAMRX_CATCH }
int iamrx_mf_setval(iamrx_mf m, double v) { IAMRX_TRY m->mf.setVal(v); IAMRX_CATCH }
int iamrx_mf_copy(iamrx_mf d, iamrx_mf s, int sc, int dc, int nc, int ng) { IAMRX_TRY MultiFab::Copy(d->mf, s->mf, sc, dc, nc, ng); IAMRX_CATCH }
int iamrx_mf_fill_boundary(iamrx_mf m, const iamrx_geom* g) { IAMRX_TRY m->mf.FillBoundary(to_geom(g)); IAMRX_CATCH }
int iamrx_mf_norm0(iamrx_mf m, int comp, int nc, int ng, double* out) { IAMRX_TRY *out = m->mf.norm0(comp, nc, ng); IAMRX_CATCH }

static AbecCoef make_coef(double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz, int tensor)
{
    AbecCoef c;
    c.alpha = alpha; c.beta = beta; c.a = a ? &a->mf : nullptr;
    c.b[0] = &bx->mf; c.b[1] = &by->mf; c.b[2] = &bz->mf; c.tensor = tensor;
    return c;
}

int iamrx_abec_gsrb(const iamrx_geom* g, double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                    iamrx_mf phi, iamrx_mf rhs, int redblack, double omega, const int lobc[3], const int hibc[3], int maxorder)
{
    IAMRX_TRY
    DomainBC b = to_bc(lobc, hibc, maxorder);
    abec_gsrb(to_geom(g), make_coef(alpha, beta, a, bx, by, bz, 0), phi->mf, rhs->mf, redblack, omega, &b, 1);
    IAMRX_CATCH
}

// One operation of the finest-level kernel FORMS the cell-centred multigrid selects by itself inside a solve (k_abec_gsrb2, k_abec_resid_restrict),
// on caller data, so that they can be compared with the oracle directly at sizes that reach their multi-workgroup / plane-marching paths
int iamrx_abec_form(const iamrx_geom* g, int coef, iamrx_mf rho, int rho_comp, double scale, const double bu[3], double beta, int op,
                    iamrx_mf phi, iamrx_mf rhs, iamrx_mf out, double omega, const int lobc[3], const int hibc[3], int maxorder)
{
    IAMRX_TRY
    Geometry gg = to_geom(g);
    DomainBC b = to_bc(lobc, hibc, maxorder);
    MultiFab bf[3];
    MultiFab* bp[3];
    for (int d = 0; d < 3; ++d) { bf[d].define(phi->mf.layout, face_type(d), 1, 0); bp[d] = &bf[d]; }
    AbecCoef c;
    c.alpha = 0.0; c.beta = beta; c.a = nullptr; c.tensor = 0;
    for (int d = 0; d < 3; ++d) c.b[d] = bp[d];
    if (coef == 1) {
        if (!rho || rho->mf.ngrow < 1) throw Error("iamrx_abec_form: coef 1 needs rho with a filled ghost cell");
        if (op != 8 && op != 9) mac_bcoef(bp, rho->mf, rho_comp, scale);
        c.sig = &rho->mf; c.sig_comp = rho_comp; c.sig_scale = scale;
    } else if (coef == 2) {
        for (int d = 0; d < 3; ++d) { bf[d].setVal(bu[d]); c.bu[d] = bu[d]; }
        c.b_uniform = 1;
    } else throw Error("iamrx_abec_form: coef is 1 (recomputed from rho) or 2 (uniform)");
    const bool wrap = op == 4 || op == 5;
    if (wrap && !periodic_wrap_ok(gg, *phi->mf.layout, 1)) throw Error("iamrx_abec_form: op 4 / 5 need one box spanning a periodic domain");
    if (op == 0 || op == 1 || wrap) abec_gsrb(gg, c, phi->mf, rhs->mf, op & 1, omega, &b, 1, false, wrap);
    else if (op == 2) abec_residual(gg, c, out->mf, phi->mf, &rhs->mf);
    else if (op == 3) {
        if (!abec_resid_restrict_ok(c, phi->mf, rhs->mf)) throw Error("iamrx_abec_form: the fused residual + restriction does not apply to these arrays");
        abec_resid_restrict(gg, c, out->mf, phi->mf, rhs->mf);
    } else if (op == 6 || op == 7) {
        if (!abec_gsrb_rb_ok(gg, c, phi->mf, 1, &b)) throw Error("iamrx_abec_form: the one-launch red + black sweep does not apply to this level");
        abec_gsrb_rb(gg, c, phi->mf, out->mf, rhs->mf, omega, op == 7, &b, 1);
    } else if (op == 8 || op == 9) {
        // the same sweep on a level of several boxes (k_abec_rb_ghost + k_abec_gsrb_rb<.., NBR>): phi and out with two ghost layers, rhs with
        // one, rho with two (its ghost cells beyond domain walls: the caller's); the ghost fills of the sweep are done here
        MultiFab& p = phi->mf;
        if (!abec_gsrb_rb_nbr_ok(gg, c, p, rhs->mf, 1, &b)) throw Error("iamrx_abec_form: the multi-box red + black sweep does not apply to this level / these arrays");
        if (coef == 1) rho->mf.FillBoundary(gg);
        rhs->mf.FillBoundary(gg);
        if (op == 8) p.FillBoundary(gg);
        abec_gsrb_rb_nbr(gg, c, p, out->mf, rhs->mf, omega, op == 9, &b, 1);
    } else if (op == 10 || op == 11) {
        // the sweep on a refined box strictly inside the domain (ratio 2): every face a coarse/fine face whose homogeneous ghost value of
        // order `maxorder` the kernel forms itself (k_abec_gsrb_rb<.., W3>); phi's ghost cells are not read
        if (!abec_gsrb_rb_cf_ok(gg, c, phi->mf)) throw Error("iamrx_abec_form: the coarse/fine red + black sweep does not apply to this level");
        double loc[3];
        for (int d = 0; d < 3; ++d) loc[d] = 0.5 * 2 * gg.dx[d];
        const CfTab tab = cf_make_tab(loc, gg.dx, maxorder);
        abec_gsrb_rb(gg, c, phi->mf, out->mf, rhs->mf, omega, op == 11, &b, 1, &tab);
    } else throw Error("iamrx_abec_form: bad op");
    IAMRX_CATCH
}

// one red+black sweep incl. the BC / ghost fills in front of each colour (fused = 1: the single-pass out-of-place kernel + the
// black pass over the box surfaces); homogeneous BC as inside a V-cycle
int iamrx_abec_gsrb_sweep(const iamrx_geom* g, double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                          iamrx_mf phi, iamrx_mf rhs, double omega, const int lobc[3], const int hibc[3], int maxorder, int fused)
{
    IAMRX_TRY
    Geometry gg = to_geom(g);
    DomainBC b = to_bc(lobc, hibc, maxorder);
    AbecCoef c = make_coef(alpha, beta, a, bx, by, bz, 0);
    MultiFab& p = phi->mf;
    auto fill = [&](MultiFab& m) { m.FillBoundary(gg); abec_apply_domain_bc(gg, m, b, false, nullptr); };
    if (fused) {
        MultiFab buf(p.layout, cell_type(), p.ncomp, 1);
        fill(p);
        abec_gsrb_fused(gg, c, p, buf, rhs->mf, omega, &b, 1);
        fill(buf);
        abec_gsrb(gg, c, buf, rhs->mf, 1, omega, &b, 1, true);
        MultiFab::Copy(p, buf, 0, 0, p.ncomp, 0);
    } else {
        for (int rb = 0; rb < 2; ++rb) { fill(p); abec_gsrb(gg, c, p, rhs->mf, rb, omega, &b, 1); }
    }
    IAMRX_CATCH
}

int iamrx_abec_residual(const iamrx_geom* g, double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                        iamrx_mf out, iamrx_mf phi, iamrx_mf rhs, int tensor)
{
    IAMRX_TRY
    abec_residual(to_geom(g), make_coef(alpha, beta, a, bx, by, bz, tensor), out->mf, phi->mf, rhs ? &rhs->mf : nullptr);
    IAMRX_CATCH
}

int iamrx_cc_restrict(iamrx_mf c, iamrx_mf f) { IAMRX_TRY cc_restrict(c->mf, f->mf); IAMRX_CATCH }
int iamrx_cc_prolong_add(iamrx_mf f, iamrx_mf c) { IAMRX_TRY cc_prolong_add(f->mf, c->mf); IAMRX_CATCH }
int iamrx_face_avgdown(iamrx_mf c, iamrx_mf f, int dir) { IAMRX_TRY face_avgdown(c->mf, f->mf, dir); IAMRX_CATCH }

int iamrx_abec_solve_cf(const iamrx_geom* g, double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                        iamrx_mf phi, iamrx_mf rhs, const int lobc[3], const int hibc[3], iamrx_mf crse_phi, const iamrx_geom* cgeom,
                        int ratio, double rel_tol, double abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    IAMRX_ASSERT(ratio == 2);
    Geometry gg = to_geom(g);
    MGOpts op = to_opts(o);
    CellMG mg(gg, phi->mf.layout, phi->mf.ncomp, to_bc(lobc, hibc, op.maxorder), op);
    mg.setScalars(alpha, beta);
    if (a) mg.setACoeffs(&a->mf);
    const MultiFab* b[3] = {&bx->mf, &by->mf, &bz->mf};
    mg.setBCoeffs(b);
    mg.setCoarseFineBC(crse_phi ? &crse_phi->mf : nullptr, to_geom(cgeom), ratio);
    mg.prepare();
    MGStats s = mg.solve(phi->mf, rhs->mf, rel_tol, abs_tol);
    from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_abec_solve(const iamrx_geom* g, double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                     iamrx_mf phi, iamrx_mf rhs, const int lobc[3], const int hibc[3], double rtol, double atol,
                     const iamrx_mg_opts* o, int tensor, iamrx_mg_stats* st)
{
    IAMRX_TRY
    Geometry gg = to_geom(g);
    MGOpts op = to_opts(o);
    CellMG mg(gg, phi->mf.layout, phi->mf.ncomp, to_bc(lobc, hibc, op.maxorder), op);
    mg.setScalars(alpha, beta);
    if (a) mg.setACoeffs(&a->mf);
    const MultiFab* b[3] = {&bx->mf, &by->mf, &bz->mf};
    MultiFab tb[3];
    if (tensor) {
        // MLTensorOp::setShearViscosity: b_d(comp) = eta_d * (comp == d ? 4/3 : 1)
        for (int d = 0; d < 3; ++d) {
            tb[d].define(phi->mf.layout, face_type(d), 3, 0);
            tensor_bcoef(tb[d], *b[d], d);
            b[d] = &tb[d];
        }
        mg.setTensor(true);
    }
    mg.setBCoeffs(b);
    mg.prepare();
    MGStats s = mg.solve(phi->mf, rhs->mf, rtol, atol);
    from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_mlmg_mac_solve(const iamrx_geom* g, iamrx_mf ux, iamrx_mf uy, iamrx_mf uz, iamrx_mf rho, int rho_comp, iamrx_mf S,
                         iamrx_mf mac_phi, double rhs_scale, const int lobc[3], const int hibc[3], double mac_tol,
                         double mac_abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    MultiFab* um[3] = {&ux->mf, &uy->mf, &uz->mf};
    MGStats s = mlmg_mac_solve(to_geom(g), um, rho->mf, rho_comp, S ? &S->mf : nullptr, mac_phi->mf, rhs_scale,
                               to_bc(lobc, hibc, op.maxorder), mac_tol, mac_abs_tol, op, nullptr);
    from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_mlmg_mac_solve_cf(const iamrx_geom* g, iamrx_mf ux, iamrx_mf uy, iamrx_mf uz, iamrx_mf rho, int rho_comp, iamrx_mf S,
                            iamrx_mf mac_phi, double rhs_scale, const int lobc[3], const int hibc[3], iamrx_mf crse_phi,
                            const iamrx_geom* cgeom, int ratio, double mac_tol, double mac_abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    IAMRX_ASSERT(ratio == 2 && cgeom);
    MGOpts op = to_opts(o);
    MultiFab* um[3] = {&ux->mf, &uy->mf, &uz->mf};
    const Geometry cg = to_geom(cgeom);
    MGStats s = mlmg_mac_solve(to_geom(g), um, rho->mf, rho_comp, S ? &S->mf : nullptr, mac_phi->mf, rhs_scale,
                               to_bc(lobc, hibc, op.maxorder), mac_tol, mac_abs_tol, op, nullptr, crse_phi ? &crse_phi->mf : nullptr, &cg, ratio);
    from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_derive_mag_vort(const iamrx_geom* g, iamrx_mf out, int ocomp, iamrx_mf vel, int vcomp)
{
    IAMRX_TRY derive_mag_vort(to_geom(g), out->mf, ocomp, vel->mf, vcomp); IAMRX_CATCH
}

int iamrx_error_tag(const iamrx_geom* g, iamrx_mf tags, iamrx_mf field, int comp, int mode, double value, int level,
                    const double* realbox_lo, const double* realbox_hi)
{
    IAMRX_TRY error_tag(to_geom(g), tags->mf, field->mf, comp, mode, value, level, realbox_lo, realbox_hi); IAMRX_CATCH
}

// the host side of the grid generation on host arrays (no device needed: tests of the regrid's host logic)
int iamrx_host_cluster_tags(const unsigned char* tags, const int dom_lo[3], const int dom_hi[3], int blocking_factor, int max_grid_size, double grid_eff,
                            int n_error_buf, const unsigned char* allowed, int* boxes, int* nboxes)
{
    IAMRX_TRY
    BoxD dom;
    for (int d = 0; d < 3; ++d) { dom.lo[d] = dom_lo[d]; dom.hi[d] = dom_hi[d]; }
    std::vector<BoxD> bx = cluster_tags(tags, dom, blocking_factor, max_grid_size, grid_eff, n_error_buf, nullptr, allowed);
    if ((int)bx.size() > *nboxes) throw Error("iamrx_host_cluster_tags: box capacity too small");
    *nboxes = (int)bx.size();
    for (size_t q = 0; q < bx.size(); ++q) for (int d = 0; d < 3; ++d) { boxes[6 * q + d] = bx[q].lo[d]; boxes[6 * q + 3 + d] = bx[q].hi[d]; }
    IAMRX_CATCH
}
int iamrx_host_erode(unsigned char* map, const int n[3], const int periodic[3], int passes)
{
    IAMRX_TRY
    std::vector<unsigned char> m(map, map + (size_t)n[0] * n[1] * n[2]);
    int lo[3] = {n[0], n[1], n[2]}, hi[3] = {-1, -1, -1};
    for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
        if (m[((size_t)k * n[1] + j) * n[0] + i]) { const int c[3] = {i, j, k}; for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], c[d]); hi[d] = std::max(hi[d], c[d]); } }
    erode_map(m, n, periodic, passes, lo, hi);
    std::copy(m.begin(), m.end(), map);
    IAMRX_CATCH
}
int iamrx_cluster_tags(const iamrx_geom* g, iamrx_mf tags, int blocking_factor, int max_grid_size, double grid_eff, int n_error_buf,
                       int* boxes, int* nboxes)
{
    IAMRX_TRY
    const Geometry gg = to_geom(g);
    const MultiFab& T = tags->mf;
    IAMRX_ASSERT(Context::get().comm->nranks == 1);      // tags of all boxes are gathered on the (single) rank
    const int n0 = gg.domain.len(0), n1 = gg.domain.len(1), n2 = gg.domain.len(2);
    std::vector<unsigned char> h((size_t)n0 * n1 * n2, 0);
    for (int li = 0; li < T.nlocal(); ++li) {
        const BoxD fb = T.fabbox(li);
        std::vector<double> buf((size_t)fb.npts() * T.ncomp);
        T.copy_to_host(li, buf.data());
        const BoxD vb = T.layout->lbox(li);
        for (int k = vb.lo[2]; k <= vb.hi[2]; ++k) for (int j = vb.lo[1]; j <= vb.hi[1]; ++j) for (int i = vb.lo[0]; i <= vb.hi[0]; ++i) {
            const size_t o = ((size_t)(k - fb.lo[2]) * fb.len(1) + (j - fb.lo[1])) * fb.len(0) + (i - fb.lo[0]);
            if (buf[o] != 0.0) h[((size_t)(k - gg.domain.lo[2]) * n1 + (j - gg.domain.lo[1])) * n0 + (i - gg.domain.lo[0])] = 1;
        }
    }
    std::vector<BoxD> bx = cluster_tags(h.data(), gg.domain, blocking_factor, max_grid_size, grid_eff, n_error_buf);
    if ((int)bx.size() > *nboxes) throw Error("iamrx_cluster_tags: box capacity too small");
    *nboxes = (int)bx.size();
    for (size_t q = 0; q < bx.size(); ++q) for (int d = 0; d < 3; ++d) { boxes[6 * q + d] = bx[q].lo[d]; boxes[6 * q + 3 + d] = bx[q].hi[d]; }
    IAMRX_CATCH
}

int iamrx_mac_divergence(const iamrx_geom* g, iamrx_mf div, iamrx_mf ux, iamrx_mf uy, iamrx_mf uz)
{
    IAMRX_TRY
    const MultiFab* um[3] = {&ux->mf, &uy->mf, &uz->mf};
    mac_divergence(to_geom(g), div->mf, um);
    IAMRX_CATCH
}

static std::vector<BCRec> to_bcrec(const int* b, int n)
{
    std::vector<BCRec> r(n);
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) { r[i].lo[d] = b ? b[6 * i + d] : 0; r[i].hi[d] = b ? b[6 * i + 3 + d] : 0; }
    return r;
}

int iamrx_mf_fill_physbc(iamrx_mf m, const iamrx_geom* g, int scomp, int ncomp, const int* bcrec, const double* edlo, const double* edhi)
{
    IAMRX_TRY
    auto bc = to_bcrec(bcrec, ncomp);
    fill_physbc_cc(to_geom(g), m->mf, scomp, ncomp, bc.data(), edlo, edhi);
    IAMRX_CATCH
}

int iamrx_godunov_extrap_vel_to_faces(const iamrx_geom* g, iamrx_mf vel, iamrx_mf force, iamrx_mf ux, iamrx_mf uy, iamrx_mf uz,
                                      double dt, const int* bcrec, int fit, int scheme)
{
    IAMRX_TRY
    auto bc = to_bcrec(bcrec, 3);
    MultiFab* um[3] = {&ux->mf, &uy->mf, &uz->mf};
    godunov_extrap_vel_to_faces(to_geom(g), vel->mf, force ? &force->mf : nullptr, um, dt, bc.data(), fit != 0, scheme);
    IAMRX_CATCH
}

int iamrx_godunov_compute_aofs(const iamrx_geom* g, iamrx_mf aofs, int acomp, iamrx_mf S, int ncomp, iamrx_mf force, iamrx_mf divu,
                               iamrx_mf ux, iamrx_mf uy, iamrx_mf uz, const int* iconserv, double dt, const int* bcrec,
                               int is_velocity, int fit, iamrx_mf ex, iamrx_mf ey, iamrx_mf ez, iamrx_mf fx, iamrx_mf fy, iamrx_mf fz, int scheme)
{
    IAMRX_TRY
    auto bc = to_bcrec(bcrec, ncomp);
    MultiFab* um[3] = {&ux->mf, &uy->mf, &uz->mf};
    MultiFab* ed[3] = {ex ? &ex->mf : nullptr, ey ? &ey->mf : nullptr, ez ? &ez->mf : nullptr};
    MultiFab* fl[3] = {fx ? &fx->mf : nullptr, fy ? &fy->mf : nullptr, fz ? &fz->mf : nullptr};
    godunov_compute_aofs(to_geom(g), aofs->mf, acomp, S->mf, ncomp, force ? &force->mf : nullptr, divu ? &divu->mf : nullptr, um,
                         iconserv, dt, bc.data(), is_velocity != 0, fit != 0, ex ? ed : nullptr, fx ? fl : nullptr, scheme);
    IAMRX_CATCH
}

int iamrx_nodal_residual(const iamrx_geom* g, iamrx_mf out, iamrx_mf phi, iamrx_mf sig, iamrx_mf rhs)
{
    IAMRX_TRY nodal_residual(to_geom(g), out->mf, phi->mf, sig->mf, rhs ? &rhs->mf : nullptr); IAMRX_CATCH
}
int iamrx_nodal_gs_color(const iamrx_geom* g, iamrx_mf phi, iamrx_mf rhs, iamrx_mf sig, int color)
{
    IAMRX_TRY nodal_gs_color(to_geom(g), phi->mf, rhs->mf, sig->mf, color); IAMRX_CATCH
}
int iamrx_nodal_gs_sweep(const iamrx_geom* g, iamrx_mf phi, iamrx_mf rhs, iamrx_mf sig, int fused)
{
    IAMRX_TRY
    Geometry gg = to_geom(g);
    if (fused == 3 || fused == 4) {
        // timing aid: the two smoother launches of a sweep alone (no ghost fills, result left in a scratch buffer); 4: with the index wrap
        // of a single box that spans a periodic domain
        // (scratch kept between calls of a timing loop; heap object never destroyed: no device free from a static destructor after HIP is gone)
        static MultiFab& xb = *new MultiFab();
        if (!xb.defined() || xb.layout != phi->mf.layout || xb.ngrow != phi->mf.ngrow) xb.define(phi->mf.layout, node_type(), 1, phi->mf.ngrow);
        const bool wrap = fused == 4;
        if (wrap && !periodic_wrap_ok(gg, *phi->mf.layout, 4)) throw Error("iamrx_nodal_gs_sweep(4): not a single box spanning a periodic domain");
        // IAMRX_BENCH_CSIG = c != 0: the constant-sigma variant with sigma = c (the caller's sigma array holds that constant)
        const double cs = tune("BENCH_CSIG", 0.0);
        nodal_gs_fused_pass(gg, phi->mf, phi->mf, xb, rhs->mf, sig->mf, 0, wrap, nullptr, cs != 0.0 ? &cs : nullptr);
        nodal_gs_fused_pass(gg, phi->mf, xb, xb, rhs->mf, sig->mf, 1, wrap, nullptr, cs != 0.0 ? &cs : nullptr);
    } else if (fused == 2) {
        if (!nodal_smooth_small(gg, phi->mf, rhs->mf, sig->mf, 1)) throw Error("level does not qualify for the single-workgroup smoother");
        phi->mf.FillBoundary(gg);
    } else if (fused) {
        MultiFab xb(phi->mf.layout, node_type(), 1, phi->mf.ngrow);
        phi->mf.FillBoundary(gg);
        nodal_gs_fused_pass(gg, phi->mf, phi->mf, xb, rhs->mf, sig->mf, 0);
        xb.FillBoundary(gg);
        nodal_gs_fused_pass(gg, phi->mf, xb, xb, rhs->mf, sig->mf, 1);
        MultiFab::Copy(phi->mf, xb, 0, 0, 1, 0);
    } else {
        for (int c = 0; c < 8; ++c) { phi->mf.FillBoundary(gg); nodal_gs_color(gg, phi->mf, rhs->mf, sig->mf, c); }
    }
    IAMRX_CATCH
}
int iamrx_nodal_restrict(iamrx_mf c, iamrx_mf f) { IAMRX_TRY nodal_restrict(c->mf, f->mf); IAMRX_CATCH }
int iamrx_nodal_interp_add(iamrx_mf f, iamrx_mf c, iamrx_mf s) { IAMRX_TRY nodal_interp_add(f->mf, c->mf, s->mf); IAMRX_CATCH }
int iamrx_nodal_divu(const iamrx_geom* g, iamrx_mf rhs, iamrx_mf vel, int vcomp)
{
    IAMRX_TRY
    Geometry gg = to_geom(g);
    DomainBC bc;      // non-periodic faces are treated as Neumann walls (the only non-periodic nodal BC implemented)
    for (int d = 0; d < 3; ++d) bc.lo[d] = bc.hi[d] = gg.periodic[d] ? lo_periodic : lo_neumann;
    bc.maxorder = 2;
    nodal_divu(gg, rhs->mf, vel->mf, vcomp, &bc);
    IAMRX_CATCH
}
int iamrx_nodal_compgrad(const iamrx_geom* g, iamrx_mf gp, iamrx_mf phi)
{
    IAMRX_TRY nodal_mknewu(to_geom(g), nullptr, 0, phi->mf, nullptr, &gp->mf, false); IAMRX_CATCH
}

int iamrx_nodal_projection(const iamrx_geom* g, iamrx_mf vel, int vcomp, iamrx_mf phi, iamrx_mf sig, int sig_comp, const int lobc[3],
                           const int hibc[3], double rel_tol, double abs_tol, const iamrx_mg_opts* o, iamrx_mf gp, int increment_gp,
                           iamrx_mg_stats* st)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    MGStats s = nodal_projection(to_geom(g), vel->mf, vcomp, phi->mf, sig->mf, sig_comp, to_bc(lobc, hibc, op.maxorder), rel_tol, abs_tol,
                                 op, gp ? &gp->mf : nullptr, increment_gp != 0);
    from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_nodal_solve(const iamrx_geom* g, iamrx_mf phi, iamrx_mf rhs, iamrx_mf sig, int sig_comp, const int lobc[3], const int hibc[3],
                      double rel_tol, double abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    NodalMG mg(to_geom(g), phi->mf.layout, to_bc(lobc, hibc, op.maxorder), op);
    mg.setSigma(sig->mf, sig_comp);
    MGStats s = mg.solve(phi->mf, rhs->mf, rel_tol, abs_tol);
    from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_tensor_apply(const iamrx_geom* g, iamrx_mf out, iamrx_mf vel, double a, double b, iamrx_mf acoef, iamrx_mf ex, iamrx_mf ey,
                       iamrx_mf ez, const int* lobc, const int* hibc, int nbc, int maxorder)
{
    IAMRX_TRY
    IAMRX_ASSERT(nbc == 1 || nbc == 3);
    const MultiFab* eta[3] = {&ex->mf, &ey->mf, &ez->mf};
    DomainBC bcs[3];
    for (int n = 0; n < nbc; ++n) bcs[n] = to_bc(lobc + 3 * n, hibc + 3 * n, maxorder);
    tensor_apply(to_geom(g), out->mf, vel->mf, a, b, acoef ? &acoef->mf : nullptr, eta, bcs, nbc);
    IAMRX_CATCH
}

int iamrx_tensor_solve(const iamrx_geom* g, iamrx_mf soln, iamrx_mf rhs, double a, double b, iamrx_mf acoef, iamrx_mf ex, iamrx_mf ey,
                       iamrx_mf ez, const int* lobc, const int* hibc, int nbc, double tol_rel, double tol_abs, const iamrx_mg_opts* o,
                       iamrx_mg_stats* st)
{
    IAMRX_TRY
    IAMRX_ASSERT(nbc == 1 || nbc == 3);
    MGOpts op = to_opts(o);
    const MultiFab* eta[3] = {&ex->mf, &ey->mf, &ez->mf};
    DomainBC bcs[3];
    for (int n = 0; n < nbc; ++n) bcs[n] = to_bc(lobc + 3 * n, hibc + 3 * n, op.maxorder);
    MGStats s = tensor_solve(to_geom(g), soln->mf, rhs->mf, a, b, acoef ? &acoef->mf : nullptr, eta, bcs, nbc, tol_rel, tol_abs, op);
    from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_tensor_extensive_flux(const iamrx_geom* g, iamrx_mf vel, iamrx_mf ex, iamrx_mf ey, iamrx_mf ez, iamrx_mf fx, iamrx_mf fy, iamrx_mf fz,
                                double fac, int add)
{
    IAMRX_TRY
    const MultiFab* eta[3] = {&ex->mf, &ey->mf, &ez->mf};
    MultiFab* fl[3] = {&fx->mf, &fy->mf, &fz->mf};
    tensor_extensive_flux(to_geom(g), vel->mf, eta, fl, fac, add != 0);
    IAMRX_CATCH
}

int iamrx_tensor_apply_cf(const iamrx_geom* g, iamrx_mf out, iamrx_mf vel, double a, double b, iamrx_mf acoef, iamrx_mf ex, iamrx_mf ey,
                          iamrx_mf ez, const int* lobc, const int* hibc, int nbc, int maxorder, iamrx_mf crse_vel, const iamrx_geom* cgeom, int ratio)
{
    IAMRX_TRY
    IAMRX_ASSERT(nbc == 1 || nbc == 3);
    const MultiFab* eta[3] = {&ex->mf, &ey->mf, &ez->mf};
    DomainBC bcs[3];
    for (int n = 0; n < nbc; ++n) bcs[n] = to_bc(lobc + 3 * n, hibc + 3 * n, maxorder);
    Geometry cg = to_geom(cgeom);
    TensorCF cf{crse_vel ? &crse_vel->mf : nullptr, &cg, ratio};
    tensor_apply(to_geom(g), out->mf, vel->mf, a, b, acoef ? &acoef->mf : nullptr, eta, bcs, nbc, &cf);
    IAMRX_CATCH
}

int iamrx_tensor_solve_cf(const iamrx_geom* g, iamrx_mf soln, iamrx_mf rhs, double a, double b, iamrx_mf acoef, iamrx_mf ex, iamrx_mf ey,
                          iamrx_mf ez, const int* lobc, const int* hibc, int nbc, iamrx_mf crse_vel, const iamrx_geom* cgeom, int ratio,
                          double tol_rel, double tol_abs, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    IAMRX_ASSERT(nbc == 1 || nbc == 3);
    MGOpts op = to_opts(o);
    const MultiFab* eta[3] = {&ex->mf, &ey->mf, &ez->mf};
    DomainBC bcs[3];
    for (int n = 0; n < nbc; ++n) bcs[n] = to_bc(lobc + 3 * n, hibc + 3 * n, op.maxorder);
    Geometry cg = to_geom(cgeom);
    TensorCF cf{crse_vel ? &crse_vel->mf : nullptr, &cg, ratio};
    MGStats s = tensor_solve(to_geom(g), soln->mf, rhs->mf, a, b, acoef ? &acoef->mf : nullptr, eta, bcs, nbc, tol_rel, tol_abs, op, &cf);
    from_stats(s, st);
    IAMRX_CATCH
}

// ---- Diffusion operator entries on caller-owned data (diffusion.hip) -----------------------------------------------------------------
namespace {
struct CrseArgs { DiffusionCrse dc; Geometry cg; bool on = false; };
void to_crse(const iamrx_diffusion_crse* c, CrseArgs& a)
{
    if (!c) return;
    a.cg = to_geom(c->cgeom);
    a.dc = DiffusionCrse{c->crse_old ? &c->crse_old->mf : nullptr, c->crse_new ? &c->crse_new->mf : nullptr, &a.cg, c->ratio};
    a.on = true;
}
}  // namespace

int iamrx_diffuse_scalar(const iamrx_geom* g, iamrx_mf S_old, iamrx_mf Rho_old, iamrx_mf S_new, iamrx_mf Rho_new, int sigma, int rho_comp, double dt,
                         double be_cn_theta, iamrx_mf rho_half, int rho_flag, const iamrx_mf* fluxn, const iamrx_mf* fluxnp1, iamrx_mf delta_rhs,
                         int rhs_comp, const iamrx_mf* betan, const iamrx_mf* betanp1, const int* lobc, const int* hibc,
                         const iamrx_diffusion_crse* crse, int add_old_time_divFlux, double visc_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    MultiFab *fn[3] = {nullptr, nullptr, nullptr}, *fp[3] = {nullptr, nullptr, nullptr};
    const MultiFab *bn[3] = {nullptr, nullptr, nullptr}, *bp[3];
    for (int d = 0; d < 3; ++d) {
        if (fluxn && fluxnp1) { fn[d] = &fluxn[d]->mf; fp[d] = &fluxnp1[d]->mf; }
        if (betan) bn[d] = &betan[d]->mf;
        bp[d] = &betanp1[d]->mf;
    }
    CrseArgs ca;
    to_crse(crse, ca);
    MGStats s = diffuse_scalar(to_geom(g), S_old ? &S_old->mf : nullptr, Rho_old ? &Rho_old->mf : nullptr, S_new->mf, Rho_new ? &Rho_new->mf : nullptr, sigma,
                               rho_comp, dt, be_cn_theta, rho_half->mf, rho_flag, (fluxn && fluxnp1) ? fn : nullptr, (fluxn && fluxnp1) ? fp : nullptr,
                               delta_rhs ? &delta_rhs->mf : nullptr, rhs_comp, betan ? bn : nullptr, bp, to_bc(lobc, hibc, 2), ca.on ? &ca.dc : nullptr,
                               add_old_time_divFlux != 0, visc_tol, op);
    if (st) from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_diffuse_tensor_velocity(const iamrx_geom* g, iamrx_mf U_old, iamrx_mf U_new, int rho_comp, double dt, double be_cn_theta, iamrx_mf rho_half,
                                  int rho_flag, iamrx_mf visc_old_term, const iamrx_mf* eta_n, const iamrx_mf* eta_np1, const int* lobc, const int* hibc,
                                  const iamrx_diffusion_crse* crse, const iamrx_mf* tflux, double visc_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st,
                                  void (*fill_new)(void* ctx, iamrx_mf U_new), void* ctx)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    const MultiFab *en[3] = {nullptr, nullptr, nullptr}, *ep[3];
    MultiFab* tf[3] = {nullptr, nullptr, nullptr};
    for (int d = 0; d < 3; ++d) { if (eta_n) en[d] = &eta_n[d]->mf; ep[d] = &eta_np1[d]->mf; if (tflux) tf[d] = &tflux[d]->mf; }
    DomainBC bcs[3];
    for (int n = 0; n < 3; ++n) bcs[n] = to_bc(lobc + 3 * n, hibc + 3 * n, 2);
    CrseArgs ca;
    to_crse(crse, ca);
    std::function<void(MultiFab&)> fill;
    if (fill_new) fill = [&](MultiFab&) { fill_new(ctx, U_new); };
    MGStats s = diffuse_tensor_velocity(to_geom(g), U_old ? &U_old->mf : nullptr, U_new->mf, rho_comp, dt, be_cn_theta, rho_half->mf, rho_flag,
                                        visc_old_term ? &visc_old_term->mf : nullptr, eta_n ? en : ep, ep, bcs, ca.on ? &ca.dc : nullptr, tflux ? tf : nullptr,
                                        visc_tol, op, fill);
    if (st) from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_diffuse_tensor_vsync(const iamrx_geom* g, iamrx_mf Vsync, double dt, double be_cn_theta, iamrx_mf rho_half, int rho_flag, iamrx_mf Rho_old,
                               iamrx_mf Rho_new, int rho_comp, const iamrx_mf* eta, const int* lobc, const int* hibc, const int* bcrec_vel,
                               const iamrx_geom* cgeom, int ratio, const iamrx_mf* tflux, double visc_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    const MultiFab* ep[3];
    MultiFab* tf[3] = {nullptr, nullptr, nullptr};
    for (int d = 0; d < 3; ++d) { ep[d] = &eta[d]->mf; if (tflux) tf[d] = &tflux[d]->mf; }
    DomainBC bcs[3];
    for (int n = 0; n < 3; ++n) bcs[n] = to_bc(lobc + 3 * n, hibc + 3 * n, 2);
    std::vector<BCRec> bv = to_bcrec(bcrec_vel, 3);
    Geometry cg;
    if (cgeom) cg = to_geom(cgeom);
    MGStats s = diffuse_tensor_Vsync(to_geom(g), Vsync->mf, dt, be_cn_theta, rho_half->mf, rho_flag, Rho_old ? &Rho_old->mf : nullptr, Rho_new ? &Rho_new->mf : nullptr,
                                     rho_comp, ep, bcs, bv.data(), cgeom ? &cg : nullptr, ratio, tflux ? tf : nullptr, visc_tol, op);
    if (st) from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_diffuse_ssync(const iamrx_geom* g, iamrx_mf Ssync, int comp, double dt, double be_cn_theta, iamrx_mf rho_half, int rho_flag, iamrx_mf Rho_new,
                        int rho_comp, const iamrx_mf* beta, const int* lobc, const int* hibc, const iamrx_geom* cgeom, int ratio, const iamrx_mf* flux,
                        double visc_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    const MultiFab* bp[3];
    MultiFab* fl[3] = {nullptr, nullptr, nullptr};
    for (int d = 0; d < 3; ++d) { bp[d] = &beta[d]->mf; if (flux) fl[d] = &flux[d]->mf; }
    Geometry cg;
    if (cgeom) cg = to_geom(cgeom);
    MGStats s = diffuse_Ssync(to_geom(g), Ssync->mf, comp, dt, be_cn_theta, rho_half->mf, rho_flag, Rho_new->mf, rho_comp, bp, to_bc(lobc, hibc, 2),
                              cgeom ? &cg : nullptr, ratio, flux ? fl : nullptr, visc_tol, op);
    if (st) from_stats(s, st);
    IAMRX_CATCH
}

struct iamrx_ns_s {
    std::unique_ptr<NavierStokes> owned;     // null for a level borrowed from an iamrx_amr hierarchy
    // borrowed handles of a hierarchy that has regridded since are kept alive as retired objects (ns_ == nullptr): every entry point
    // reports an error on them instead of touching a freed level
    struct Ref {
        NavierStokes* p = nullptr;
        NavierStokes* operator->() const { if (!p) throw Error("stale level handle: the hierarchy was regridded, fetch the level again with iamrx_amr_level"); return p; }
        NavierStokes& operator*() const { return *operator->(); }
        Ref& operator=(NavierStokes* q) { p = q; return *this; }
    } ns;
    iamrx_mf_s* views[10];
    bool probing = false;
    LayoutP layout;
};

void iamrx_ns_default_params(iamrx_ns_params* p)
{
    NSParams d;
    memset(p, 0, sizeof *p);      // fields added later start from zero for callers that fill a stack struct through this function
    p->cfl = d.cfl; p->visc_coef = d.visc_coef; p->be_cn_theta = d.be_cn_theta; p->gravity = d.gravity;
    p->mac_tol = d.mac_tol; p->mac_abs_tol = d.mac_abs_tol; p->proj_tol = d.proj_tol; p->proj_abs_tol = d.proj_abs_tol;
    p->visc_tol = d.visc_tol; p->use_forces_in_trans = d.use_forces_in_trans; p->do_mom_diff = d.do_mom_diff;
    p->init_iter = d.init_iter; p->init_vel_iter = d.init_vel_iter; p->init_shrink = d.init_shrink; p->change_max = d.change_max;
    p->fixed_dt = d.fixed_dt; p->nscal = d.nscal; p->verbose = d.verbose;
    p->init_dt = d.init_dt; p->tracer_diff_coef = d.tracer_diff_coef;
    for (int i = 0; i < 3; ++i) { p->phys_lo[i] = d.phys_lo[i]; p->phys_hi[i] = d.phys_hi[i]; }
    for (int i = 0; i < 9; ++i) { p->wall_vel_lo[i] = d.wall_vel_lo[i]; p->wall_vel_hi[i] = d.wall_vel_hi[i]; }
    for (int i = 0; i < 12; ++i) { p->scal_bc_lo[i] = d.scal_bc_lo[i]; p->scal_bc_hi[i] = d.scal_bc_hi[i]; }
    p->do_cons_trac = d.do_cons_trac;
    p->do_denminmax = d.do_denminmax; p->do_scalminmax = d.do_scalminmax;
    p->do_trac2 = d.do_trac2; p->do_cons_trac2 = d.do_cons_trac2; p->tracer2_diff_coef = d.tracer2_diff_coef; p->do_temp = d.do_temp; p->temp_cond_coef = d.temp_cond_coef;
    p->use_ppm = d.use_ppm;
}

static NSParams to_params(const iamrx_ns_params* p)
{
    NSParams q;
    q.cfl = p->cfl; q.visc_coef = p->visc_coef; q.be_cn_theta = p->be_cn_theta; q.gravity = p->gravity;
    q.mac_tol = p->mac_tol; q.mac_abs_tol = p->mac_abs_tol; q.proj_tol = p->proj_tol; q.proj_abs_tol = p->proj_abs_tol;
    q.visc_tol = p->visc_tol; q.use_forces_in_trans = p->use_forces_in_trans; q.do_mom_diff = p->do_mom_diff;
    q.init_iter = p->init_iter; q.init_vel_iter = p->init_vel_iter; q.init_shrink = p->init_shrink; q.change_max = p->change_max;
    q.fixed_dt = p->fixed_dt; q.nscal = p->nscal; q.verbose = p->verbose;
    q.init_dt = p->init_dt; q.tracer_diff_coef = p->tracer_diff_coef;
    for (int i = 0; i < 3; ++i) { q.phys_lo[i] = p->phys_lo[i]; q.phys_hi[i] = p->phys_hi[i]; }
    for (int i = 0; i < 9; ++i) { q.wall_vel_lo[i] = p->wall_vel_lo[i]; q.wall_vel_hi[i] = p->wall_vel_hi[i]; }
    for (int i = 0; i < 12; ++i) { q.scal_bc_lo[i] = p->scal_bc_lo[i]; q.scal_bc_hi[i] = p->scal_bc_hi[i]; }
    q.do_cons_trac = p->do_cons_trac;
    q.do_denminmax = p->do_denminmax; q.do_scalminmax = p->do_scalminmax;
    q.do_trac2 = p->do_trac2; q.do_cons_trac2 = p->do_cons_trac2; q.tracer2_diff_coef = p->tracer2_diff_coef; q.do_temp = p->do_temp; q.temp_cond_coef = p->temp_cond_coef;
    q.use_ppm = p->use_ppm;
    return q;
}

int iamrx_ns_create(const iamrx_geom* g, iamrx_layout l, const iamrx_ns_params* p, const iamrx_mg_opts* o, iamrx_ns* out)
{
    IAMRX_TRY
    auto* h = new iamrx_ns_s;
    h->owned = std::make_unique<NavierStokes>(to_geom(g), l->p, to_params(p), to_opts(o));
    h->ns = h->owned.get();
    h->layout = l->p;
    for (auto& v : h->views) v = nullptr;
    *out = h;
    IAMRX_CATCH
}
int iamrx_ns_destroy(iamrx_ns ns) { IAMRX_TRY delete ns; IAMRX_CATCH }
int iamrx_ns_init_rayleightaylor(iamrx_ns ns, double rho_1, double rho_2, double tra_1, double tra_2, double pertamp, double interface_width)
{
    IAMRX_TRY ns->ns->init_rayleightaylor(rho_1, rho_2, tra_1, tra_2, pertamp, interface_width); IAMRX_CATCH
}
int iamrx_ns_init_taylorgreen(iamrx_ns ns, double vfac, double a, double b, double c, double rho0)
{
    IAMRX_TRY ns->ns->init_taylorgreen(vfac, a, b, c, rho0); IAMRX_CATCH
}
int iamrx_ns_init_rest(iamrx_ns ns, double rho0) { IAMRX_TRY ns->ns->init_rest(rho0); IAMRX_CATCH }
int iamrx_ns_post_init(iamrx_ns ns, double stop_time) { IAMRX_TRY ns->ns->post_init(stop_time); IAMRX_CATCH }
int iamrx_ns_step(iamrx_ns ns, double* dt_used) { IAMRX_TRY double d = ns->ns->step(); if (dt_used) *dt_used = d; IAMRX_CATCH }
int iamrx_ns_advance(iamrx_ns ns, double dt, double* dt_est) { IAMRX_TRY double d = ns->ns->advance(dt); if (dt_est) *dt_est = d; IAMRX_CATCH }
int iamrx_ns_time(iamrx_ns ns, double* time, double* dt, int* nstep)
{
    IAMRX_TRY
    if (time) *time = ns->ns->time;
    if (dt) *dt = ns->ns->dt;
    if (nstep) *nstep = ns->ns->nstep;
    IAMRX_CATCH
}

int iamrx_ns_restart_state(iamrx_ns ns, int set, double state[16])
{
    IAMRX_TRY
    if (set) ns->ns->set_restart_state(state); else ns->ns->get_restart_state(state);
    IAMRX_CATCH
}

// non-owning view: the C handle type wraps a MultiFab by value, so expose the persistent arrays through
// a pointer-carrying subclass-free trick: a dedicated handle whose MultiFab is a shallow alias.
int iamrx_ns_data(iamrx_ns ns, int which, iamrx_mf* out)
{
    IAMRX_TRY
    NavierStokes& n = *ns->ns;
    MultiFab* m = nullptr;
    switch (which) {
    case 0: m = &n.get_new_data(0); break;
    case 1: m = &n.get_old_data(0); break;
    case 2: m = &n.get_new_data(1); break;
    case 3: m = &n.get_old_data(1); break;
    case 4: m = &n.get_new_data(2); break;
    case 5: m = &n.get_old_data(2); break;
    case 6: case 7: case 8: m = &n.umac(which - 6); break;
    case 9: m = &n.Aofs(); break;
    case 10: case 11: m = &n.mac_phi_history(which - 10); break;
    default: throw Error("iamrx_ns_data: bad selector");
    }
    // copy the current contents into a library-owned MultiFab of the same shape (old/new swap every step,
    // so a stable alias would be misleading); the caller destroys it with iamrx_mf_destroy
    auto* h = new iamrx_mf_s;
    h->mf.define(n.user_layout, m->type, m->ncomp, m->ngrow);
    relayout_copy(h->mf, *m, m->ncomp);
    *out = h;
    IAMRX_CATCH
}

// overwrite one of the level's arrays (same selectors as iamrx_ns_data) with src: problem set-up from caller data
int iamrx_ns_derive(iamrx_ns ns, const char* name, iamrx_mf out, int ocomp)
{
    IAMRX_TRY
    NavierStokes& n = *ns->ns;
    if (out->mf.layout->id == n.lay()->id) n.derive(name, out->mf, ocomp);
    else {
        IAMRX_ASSERT(out->mf.layout->id == n.user_layout->id && out->mf.type.cell());
        MultiFab t(n.lay(), cell_type(), 1, 0), u(n.user_layout, cell_type(), 1, 0);
        n.derive(name, t, 0);
        relayout_copy(u, t, 1);
        MultiFab::Copy(out->mf, u, 0, ocomp, 1, 0);
    }
    IAMRX_CATCH
}

int iamrx_ns_set_data(iamrx_ns ns, int which, iamrx_mf src)
{
    IAMRX_TRY
    NavierStokes& n = *ns->ns;
    MultiFab* m = nullptr;
    switch (which) {
    case 0: m = &n.get_new_data(0); break;
    case 1: m = &n.get_old_data(0); break;
    case 2: m = &n.get_new_data(1); break;
    case 3: m = &n.get_old_data(1); break;
    case 4: m = &n.get_new_data(2); break;
    case 5: m = &n.get_old_data(2); break;
    case 10: case 11: m = &n.mac_phi_history(which - 10); break;
    default: throw Error("iamrx_ns_set_data: bad selector");
    }
    // fewer components than the level holds: the leading ones (the state without the divu / dsdt components a temperature run appends)
    IAMRX_ASSERT(src->mf.ncomp <= m->ncomp && src->mf.ngrow == m->ngrow && (src->mf.layout->id == m->layout->id || src->mf.layout->id == n.user_layout->id));
    relayout_copy(*m, src->mf, src->mf.ncomp);
    IAMRX_CATCH
}

int iamrx_parallel_copy(iamrx_mf dst, iamrx_mf src, int scomp, int dcomp, int ncomp, int src_ng, int dst_ng, const iamrx_geom* periodic_geom)
{
    IAMRX_TRY
    Geometry g;
    if (periodic_geom) g = to_geom(periodic_geom);
    parallel_copy(dst->mf, src->mf, scomp, dcomp, ncomp, src_ng, dst_ng, periodic_geom ? &g : nullptr);
    IAMRX_CATCH
}
int iamrx_fillpatch_two_levels(iamrx_mf dst, int dcomp, double time, iamrx_mf fine_old, iamrx_mf fine_new, double t_fine_old, double t_fine_new,
                               iamrx_mf crse_old, iamrx_mf crse_new, double t_crse_old, double t_crse_new, int scomp, int ncomp,
                               const iamrx_geom* cgeom, const iamrx_geom* fgeom, int ratio, const int* bcrec, const double* edlo, const double* edhi)
{
    IAMRX_TRY
    auto bc = to_bcrec(bcrec, ncomp);
    TimeData f{fine_old ? &fine_old->mf : nullptr, &fine_new->mf, t_fine_old, t_fine_new};
    TimeData c{crse_old ? &crse_old->mf : nullptr, &crse_new->mf, t_crse_old, t_crse_new};
    fillpatch_two_levels(dst->mf, dcomp, time, f, c, scomp, ncomp, to_geom(cgeom), to_geom(fgeom), ratio, bc.data(), edlo, edhi);
    IAMRX_CATCH
}
struct iamrx_fluxreg_s { std::unique_ptr<FluxRegister> fr; };
int iamrx_fluxreg_create(iamrx_layout fine, iamrx_layout crse, const iamrx_geom* cgeom, int ratio, int ncomp, iamrx_fluxreg* out)
{
    IAMRX_TRY
    auto* h = new iamrx_fluxreg_s;
    h->fr = std::make_unique<FluxRegister>(fine->p, crse->p, to_geom(cgeom), ratio, ncomp);
    *out = h;
    IAMRX_CATCH
}
int iamrx_fluxreg_destroy(iamrx_fluxreg fr) { IAMRX_TRY delete fr; IAMRX_CATCH }
int iamrx_mac_sync_solve(const iamrx_geom* g, iamrx_fluxreg mac_reg, iamrx_mf rho_half, double dt, iamrx_layout fine, int ratio,
                         iamrx_mf ux, iamrx_mf uy, iamrx_mf uz, iamrx_mf mac_sync_phi, const int lobc[3], const int hibc[3],
                         double tol, double abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    MultiFab* uc[3] = {&ux->mf, &uy->mf, &uz->mf};
    MGStats s = mac_sync_solve(to_geom(g), *mac_reg->fr, rho_half->mf, dt, fine->p, ratio, uc, mac_sync_phi->mf, to_bc(lobc, hibc, op.maxorder),
                               tol, abs_tol, op);
    from_stats(s, st);
    IAMRX_CATCH
}
int iamrx_fluxreg_setval(iamrx_fluxreg fr, double v) { IAMRX_TRY fr->fr->setVal(v); IAMRX_CATCH }
int iamrx_fluxreg_crse_init(iamrx_fluxreg fr, iamrx_mf flux, int dir, int scomp, int dcomp, int ncomp, double mult, int add)
{
    IAMRX_TRY fr->fr->CrseInit(flux->mf, dir, scomp, dcomp, ncomp, mult, add != 0); IAMRX_CATCH
}
int iamrx_fluxreg_fine_add(iamrx_fluxreg fr, iamrx_mf flux, int dir, int scomp, int dcomp, int ncomp, double mult)
{
    IAMRX_TRY fr->fr->FineAdd(flux->mf, dir, scomp, dcomp, ncomp, mult); IAMRX_CATCH
}
int iamrx_fluxreg_reflux(iamrx_fluxreg fr, iamrx_mf S, double volume, double scale, int scomp, int dcomp, int ncomp)
{
    IAMRX_TRY fr->fr->Reflux(S->mf, volume, scale, scomp, dcomp, ncomp); IAMRX_CATCH
}
int iamrx_create_umac_grown(iamrx_mf fx, iamrx_mf fy, iamrx_mf fz, iamrx_mf cx, iamrx_mf cy, iamrx_mf cz, iamrx_mf divu,
                            const iamrx_geom* cgeom, const iamrx_geom* fgeom, int ratio)
{
    IAMRX_TRY
    MultiFab* uf[3] = {&fx->mf, &fy->mf, &fz->mf};
    const MultiFab* uc[3] = {&cx->mf, &cy->mf, &cz->mf};
    create_umac_grown(uf, uc, divu ? &divu->mf : nullptr, to_geom(cgeom), to_geom(fgeom), ratio);
    IAMRX_CATCH
}
int iamrx_average_down(iamrx_mf fine, iamrx_mf crse, int scomp, int ncomp, int ratio)
{
    IAMRX_TRY average_down(fine->mf, crse->mf, scomp, ncomp, ratio); IAMRX_CATCH
}

// ---- zero-copy alias, level projection entry, sync-operator entries
int iamrx_mf_alias(iamrx_layout l, const int type[3], int ncomp, int ngrow, double* const* dev_ptrs, iamrx_mf* out)
{
    IAMRX_TRY
    auto* h = new iamrx_mf_s;
    IndexType t{{type[0], type[1], type[2]}};
    h->mf.alias(l->p, t, ncomp, ngrow, dev_ptrs);
    *out = h;
    IAMRX_CATCH
}

int iamrx_level_project(const iamrx_geom* g, double dt, iamrx_mf U_new, int vcomp, iamrx_mf P_new, iamrx_mf Gp_old, iamrx_mf Gp_new,
                        iamrx_mf rho_half, const int lobc[3], const int hibc[3], double proj_tol, double proj_abs_tol,
                        const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    DomainBC bc;
    for (int d = 0; d < 3; ++d) { bc.lo[d] = lobc[d]; bc.hi[d] = hibc[d]; }
    bc.maxorder = 2;
    MGStats s = level_project_single(to_geom(g), dt, U_new->mf, vcomp, P_new->mf, Gp_old->mf, Gp_new->mf, rho_half->mf, bc, proj_tol, proj_abs_tol, to_opts(o));
    from_stats(s, st);
    IAMRX_CATCH
}

struct iamrx_syncreg_s { std::unique_ptr<SyncRegister> sr; };
int iamrx_syncreg_create(iamrx_layout fine, iamrx_layout crse, const iamrx_geom* cgeom, const iamrx_geom* fgeom, int ratio,
                         const int phys_lo[3], const int phys_hi[3], iamrx_syncreg* out)
{
    IAMRX_TRY
    auto* h = new iamrx_syncreg_s;
    h->sr = std::make_unique<SyncRegister>(fine->p, crse->p, to_geom(cgeom), to_geom(fgeom), ratio, phys_lo, phys_hi);
    *out = h;
    IAMRX_CATCH
}
int iamrx_syncreg_destroy(iamrx_syncreg r) { IAMRX_TRY delete r; IAMRX_CATCH }
int iamrx_syncreg_crse_init(iamrx_syncreg r, iamrx_mf sync_resid_crse, double mult) { IAMRX_TRY r->sr->CrseInit(sync_resid_crse->mf, mult); IAMRX_CATCH }
int iamrx_syncreg_fine_add(iamrx_syncreg r, iamrx_mf sync_resid_fine, double mult) { IAMRX_TRY r->sr->FineAdd(sync_resid_fine->mf, mult); IAMRX_CATCH }
int iamrx_syncreg_init_rhs(iamrx_syncreg r, iamrx_mf rhs) { IAMRX_TRY r->sr->InitRHS(rhs->mf); IAMRX_CATCH }

// a level of a multi-level nodal projection on caller-owned data.  Ghost velocities outside inflow faces: the caller's data count in full
// where the projection asks for them (inflow_scale 1: initialVelocityProject), and are zeroed for increments that carry no inflow data
// (inflow_scale 0: the sync projections, initialSyncProject; Projection::set_boundary_velocity, Source/Projection.cpp:2570-2663)
static ProjLevel to_proj_level(const iamrx_proj_level* in)
{
    ProjLevel P;
    P.g = to_geom(in->geom); P.layout = in->layout->p; P.ratio = in->ratio;
    P.nodal_bc = to_bc(in->lobc, in->hibc, 2);
    P.gp = in->gp ? &in->gp->mf : nullptr;
    const Geometry g = P.g;
    const DomainBC bc = P.nodal_bc;
    P.set_inflow = [g, bc](MultiFab& vel, double scale) {
        if (scale == 1.0) return;
        for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
            if (g.periodic[d] || (side == 0 ? bc.lo[d] : bc.hi[d]) != lo_inflow) continue;
            const int face = side == 0 ? g.domain.lo[d] - 1 : g.domain.hi[d] + 1;
            const FabD* vt = vel.d_tab;
            const int dd = d;
            const double sc = scale;
            for_each(*vel.layout, cell_type(), 1, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
                if ((dd == 0 ? i : (dd == 1 ? j : k)) == face) vt[f](i, j, k, dd) *= sc;
            });
        }
    };
    return P;
}

// Projection::MLsyncProject on caller-owned data (amrns.hip ml_sync_project)
int iamrx_mlsync_project(const iamrx_proj_level* crse, const iamrx_proj_level* fine, iamrx_mf pres_crse, iamrx_mf vel_crse, int vcomp_crse, iamrx_mf pres_fine,
                         iamrx_mf vel_fine, int vcomp_fine, iamrx_mf rho_crse, iamrx_mf rho_fine, iamrx_mf Vsync, iamrx_mf V_corr, iamrx_mf phi_crse,
                         iamrx_mf phi_fine, iamrx_syncreg rhs_sync_reg, iamrx_syncreg crse_sync_reg, double dt, int crse_iteration, int crse_dt_ratio,
                         double sync_tol, double abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    MGOpts op = to_opts(o);
    ProjLevel PL[2] = {to_proj_level(crse), to_proj_level(fine)};
    MGStats s = ml_sync_project(PL, pres_crse->mf, vel_crse->mf, vcomp_crse, pres_fine->mf, vel_fine->mf, vcomp_fine, rho_crse->mf, rho_fine->mf, Vsync->mf,
                                V_corr->mf, phi_crse->mf, phi_fine->mf, *rhs_sync_reg->sr, crse_sync_reg ? crse_sync_reg->sr.get() : nullptr, dt, crse_iteration,
                                crse_dt_ratio, sync_tol, abs_tol, op);
    if (st) from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_sync_interp(iamrx_mf fine_dst, int dcomp, iamrx_mf crse_sync, int scomp, int ncomp, const iamrx_geom* cgeom, const iamrx_geom* fgeom,
                      int ratio, const int* bcrec)
{
    IAMRX_TRY
    std::vector<BCRec> bc(ncomp);
    for (int n = 0; n < ncomp; ++n) for (int d = 0; d < 3; ++d) { bc[n].lo[d] = bcrec ? bcrec[6 * n + d] : 0; bc[n].hi[d] = bcrec ? bcrec[6 * n + 3 + d] : 0; }
    sync_interp_cellcons(fine_dst->mf, dcomp, crse_sync->mf, scomp, ncomp, to_geom(cgeom), to_geom(fgeom), ratio, bc.data());
    IAMRX_CATCH
}

int iamrx_godunov_compute_aofs_sync(const iamrx_geom* g, iamrx_mf sync, int acomp, iamrx_mf S, int ncomp, iamrx_mf force, iamrx_mf divu,
                                    iamrx_mf umac_x, iamrx_mf umac_y, iamrx_mf umac_z, iamrx_mf ucorr_x, iamrx_mf ucorr_y, iamrx_mf ucorr_z,
                                    const int* iconserv, double dt, const int* bcrec, int is_velocity, int use_forces_in_trans,
                                    iamrx_mf flux_x, iamrx_mf flux_y, iamrx_mf flux_z, int scheme)
{
    IAMRX_TRY
    std::vector<BCRec> bc(ncomp);
    for (int n = 0; n < ncomp; ++n) for (int d = 0; d < 3; ++d) { bc[n].lo[d] = bcrec ? bcrec[6 * n + d] : 0; bc[n].hi[d] = bcrec ? bcrec[6 * n + 3 + d] : 0; }
    MultiFab* um[3] = {&umac_x->mf, &umac_y->mf, &umac_z->mf};
    MultiFab* uc[3] = {&ucorr_x->mf, &ucorr_y->mf, &ucorr_z->mf};
    MultiFab* fl[3] = {flux_x ? &flux_x->mf : nullptr, flux_y ? &flux_y->mf : nullptr, flux_z ? &flux_z->mf : nullptr};
    godunov_compute_aofs_sync(to_geom(g), sync->mf, acomp, S->mf, ncomp, force ? &force->mf : nullptr, divu ? &divu->mf : nullptr, um, uc,
                              iconserv, dt, bc.data(), is_velocity != 0, use_forces_in_trans != 0, flux_x ? fl : nullptr, scheme);
    IAMRX_CATCH
}

int iamrx_syncreg_comp_add(iamrx_syncreg r, iamrx_mf sync_resid_fine, const iamrx_geom* fgeom, iamrx_layout finer, int finer_ratio, double mult)
{
    IAMRX_TRY
    r->sr->CompAdd(sync_resid_fine->mf, to_geom(fgeom), finer->p, finer_ratio, mult);
    IAMRX_CATCH
}

int iamrx_mac_sync_compute(const iamrx_geom* g, iamrx_mf ucorr_x, iamrx_mf ucorr_y, iamrx_mf ucorr_z, iamrx_mf Vsync, iamrx_mf Ssync, iamrx_mf S_vel,
                           iamrx_mf S_scal, int nscal, iamrx_mf visc_vel, iamrx_mf tforce_scal, iamrx_mf gradp, iamrx_mf divu, iamrx_mf umac_x,
                           iamrx_mf umac_y, iamrx_mf umac_z, const int* iconserv_scal, int do_mom_diff, double gravity, double dt, const int* bcrec_vel,
                           const int* bcrec_scal, int use_forces_in_trans, int scheme, iamrx_mf fluxv_x, iamrx_mf fluxv_y, iamrx_mf fluxv_z,
                           iamrx_mf fluxs_x, iamrx_mf fluxs_y, iamrx_mf fluxs_z)
{
    IAMRX_TRY
    std::vector<BCRec> bv(3), bs(nscal);
    for (int n = 0; n < 3; ++n) for (int d = 0; d < 3; ++d) { bv[n].lo[d] = bcrec_vel ? bcrec_vel[6 * n + d] : 0; bv[n].hi[d] = bcrec_vel ? bcrec_vel[6 * n + 3 + d] : 0; }
    for (int n = 0; n < nscal; ++n) for (int d = 0; d < 3; ++d) { bs[n].lo[d] = bcrec_scal ? bcrec_scal[6 * n + d] : 0; bs[n].hi[d] = bcrec_scal ? bcrec_scal[6 * n + 3 + d] : 0; }
    MultiFab* um[3] = {&umac_x->mf, &umac_y->mf, &umac_z->mf};
    MultiFab* uc[3] = {&ucorr_x->mf, &ucorr_y->mf, &ucorr_z->mf};
    MultiFab* fv[3] = {fluxv_x ? &fluxv_x->mf : nullptr, fluxv_y ? &fluxv_y->mf : nullptr, fluxv_z ? &fluxv_z->mf : nullptr};
    MultiFab* fs[3] = {fluxs_x ? &fluxs_x->mf : nullptr, fluxs_y ? &fluxs_y->mf : nullptr, fluxs_z ? &fluxs_z->mf : nullptr};
    mac_sync_compute(to_geom(g), uc, Vsync->mf, Ssync->mf, S_vel->mf, S_scal->mf, nscal, visc_vel ? &visc_vel->mf : nullptr, tforce_scal ? &tforce_scal->mf : nullptr,
                     gradp->mf, divu ? &divu->mf : nullptr, um, iconserv_scal, do_mom_diff != 0, gravity, dt, bv.data(), bs.data(), use_forces_in_trans != 0, scheme,
                     fluxv_x ? fv : nullptr, fluxs_x ? fs : nullptr);
    IAMRX_CATCH
}

int iamrx_mac_sync_compute_edge(const iamrx_geom* g, iamrx_mf ucorr_x, iamrx_mf ucorr_y, iamrx_mf ucorr_z, iamrx_mf Sync, int sync_indx, iamrx_mf edge_x,
                                iamrx_mf edge_y, iamrx_mf edge_z, int edge_comp, iamrx_mf flux_x, iamrx_mf flux_y, iamrx_mf flux_z)
{
    IAMRX_TRY
    MultiFab* uc[3] = {&ucorr_x->mf, &ucorr_y->mf, &ucorr_z->mf};
    MultiFab* ed[3] = {&edge_x->mf, &edge_y->mf, &edge_z->mf};
    MultiFab* fl[3] = {flux_x ? &flux_x->mf : nullptr, flux_y ? &flux_y->mf : nullptr, flux_z ? &flux_z->mf : nullptr};
    mac_sync_compute_edge(to_geom(g), uc, Sync->mf, sync_indx, ed, edge_comp, flux_x ? fl : nullptr);
    IAMRX_CATCH
}

int iamrx_initial_velocity_project(int nlev, const iamrx_proj_level* levels, const iamrx_mf* vel, const int* vcomp, const iamrx_mf* pres, const iamrx_mf* rho,
                                   const int* rho_comp, const iamrx_mf* divu, const int* divu_comp, double proj_tol, double proj_abs_tol,
                                   const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    std::vector<ProjLevel> PL(nlev);
    std::vector<MultiFab*> v(nlev), p(nlev);
    std::vector<const MultiFab*> r(nlev, nullptr), dv(nlev, nullptr);
    for (int l = 0; l < nlev; ++l) {
        PL[l] = to_proj_level(&levels[l]);
        v[l] = &vel[l]->mf; p[l] = &pres[l]->mf;
        if (rho && rho[l]) r[l] = &rho[l]->mf;
        if (divu && divu[l]) dv[l] = &divu[l]->mf;
    }
    MGStats s = initial_velocity_project(PL, v.data(), vcomp, p.data(), rho ? r.data() : nullptr, rho_comp, divu ? dv.data() : nullptr, divu_comp, proj_tol,
                                         proj_abs_tol, to_opts(o));
    if (st) from_stats(s, st);
    IAMRX_CATCH
}

int iamrx_initial_sync_project(int nlev, const iamrx_proj_level* levels, const iamrx_mf* vel_new, const int* vcomp, const iamrx_mf* vel_old, const iamrx_mf* phi,
                               const iamrx_mf* pres_new, const iamrx_mf* rho_half, const iamrx_mf* divu_new, const iamrx_mf* divu_old, const int* divu_comp,
                               double dt, double proj_tol, double proj_abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st)
{
    IAMRX_TRY
    std::vector<ProjLevel> PL(nlev);
    std::vector<MultiFab*> vn(nlev), ph(nlev), pn(nlev, nullptr);
    std::vector<const MultiFab*> vo(nlev), rh(nlev), dn(nlev, nullptr), dol(nlev, nullptr);
    for (int l = 0; l < nlev; ++l) {
        PL[l] = to_proj_level(&levels[l]);
        vn[l] = &vel_new[l]->mf; vo[l] = &vel_old[l]->mf; ph[l] = &phi[l]->mf; rh[l] = &rho_half[l]->mf;
        if (pres_new && pres_new[l]) pn[l] = &pres_new[l]->mf;
        if (divu_new && divu_new[l]) dn[l] = &divu_new[l]->mf;
        if (divu_old && divu_old[l]) dol[l] = &divu_old[l]->mf;
    }
    MGStats s = initial_sync_project(PL, vn.data(), vcomp, vo.data(), ph.data(), pres_new ? pn.data() : nullptr, rh.data(), divu_new ? dn.data() : nullptr,
                                     divu_old ? dol.data() : nullptr, divu_comp, dt, proj_tol, proj_abs_tol, to_opts(o));
    if (st) from_stats(s, st);
    IAMRX_CATCH
}

// ---- AMR hierarchy (amrns.hip)
struct iamrx_amr_s {
    std::unique_ptr<AmrNS> amr;
    std::vector<std::unique_ptr<iamrx_ns_s>> levels;     // borrowed level handles
    std::vector<std::unique_ptr<iamrx_ns_s>> retired;    // handles given out before a regrid: alive until the hierarchy dies, ns == nullptr
};
int iamrx_amr_create(const iamrx_geom* g0, int nlev, const iamrx_layout* layouts, int ratio, const iamrx_ns_params* p, const iamrx_mg_opts* o, iamrx_amr* out)
{
    IAMRX_TRY
    std::vector<LayoutP> ls;
    for (int l = 0; l < nlev; ++l) ls.push_back(layouts[l]->p);
    auto* h = new iamrx_amr_s;
    h->amr = std::make_unique<AmrNS>(to_geom(g0), ls, ratio, to_params(p), to_opts(o));
    for (int l = 0; l < nlev; ++l) {
        auto v = std::make_unique<iamrx_ns_s>();
        v->ns = &h->amr->level(l);
        v->layout = ls[l];
        for (auto& q : v->views) q = nullptr;
        h->levels.push_back(std::move(v));
    }
    *out = h;
    IAMRX_CATCH
}
int iamrx_amr_destroy(iamrx_amr a) { IAMRX_TRY delete a; IAMRX_CATCH }
static void amr_refresh_levels(iamrx_amr a)
{
    // the level handles given out so far stay valid objects (retired: every iamrx_ns_* entry returns an error on them) but the
    // levels they referred to no longer exist: rebuild the table
    for (auto& v : a->levels) {
        v->ns = nullptr;
        for (auto& q : v->views) { delete q; q = nullptr; }
        v->layout.reset();
        a->retired.push_back(std::move(v));
    }
    a->levels.clear();
    for (int l = 0; l < a->amr->nlevels(); ++l) {
        auto v = std::make_unique<iamrx_ns_s>();
        v->ns = &a->amr->level(l);
        v->layout = a->amr->level(l).user_layout;
        for (auto& q : v->views) q = nullptr;
        a->levels.push_back(std::move(v));
    }
}
int iamrx_amr_set_regrid(iamrx_amr a, int max_level, int regrid_int, int blocking_factor, int max_grid_size, double grid_eff, int n_error_buf,
                         int nrules, const iamrx_tag_rule* rules)
{
    IAMRX_TRY
    AmrNS::RegridOpts r;
    r.max_level = max_level; r.regrid_int = regrid_int; r.blocking_factor = blocking_factor; r.max_grid_size = max_grid_size;
    r.grid_eff = grid_eff; r.n_error_buf = n_error_buf;
    for (int q = 0; q < nrules; ++q) {
        AmrNS::TagRule t;
        t.comp = rules[q].comp; t.mode = rules[q].mode; t.max_level = rules[q].max_level;
        for (int v = 0; v < rules[q].nvalue && v < 8; ++v) t.value.push_back(rules[q].value[v]);
        t.has_box = rules[q].has_box != 0;
        for (int d = 0; d < 3; ++d) { t.box_lo[d] = rules[q].box_lo[d]; t.box_hi[d] = rules[q].box_hi[d]; }
        r.rules.push_back(t);
    }
    a->amr->set_regrid(r);
    IAMRX_CATCH
}
int iamrx_amr_regrid_log_count(iamrx_amr a, int* nevents) { IAMRX_TRY *nevents = (int)a->amr->regrid_log.size(); IAMRX_CATCH }
int iamrx_amr_regrid_log_event(iamrx_amr a, int event, int* lbase, double* time, int* nlevels, int* nboxes, int* boxes)
{
    IAMRX_TRY
    const auto& log = a->amr->regrid_log;
    if (event < 0 || event >= (int)log.size()) throw Error("iamrx_amr_regrid_log_event: no such event");
    const auto& e = log[event];
    *lbase = e.lbase; *time = e.time; *nlevels = (int)e.grids.size();
    int q = 0;
    for (size_t l = 0; l < e.grids.size(); ++l) {
        if (nboxes) nboxes[l] = (int)e.grids[l].size();
        if (boxes) for (const BoxD& b : e.grids[l]) { for (int d = 0; d < 3; ++d) { boxes[q + d] = b.lo[d]; boxes[q + 3 + d] = b.hi[d]; } q += 6; }
    }
    IAMRX_CATCH
}
int iamrx_amr_set_compute_new_dt_on_regrid(iamrx_amr a, int on) { IAMRX_TRY a->amr->set_compute_new_dt_on_regrid(on != 0); IAMRX_CATCH }
int iamrx_amr_set_outflow_tagging(iamrx_amr a, int do_refine_outflow, int do_derefine_outflow, int nbuf_outflow)
{
    IAMRX_TRY
    if (do_refine_outflow && do_derefine_outflow) throw Error("iamrx_amr_set_outflow_tagging: do_refine_outflow and do_derefine_outflow cannot both be set (NavierStokesBase.cpp:514-516)");
    a->amr->set_outflow_tagging(do_refine_outflow, do_derefine_outflow, nbuf_outflow);
    IAMRX_CATCH
}
int iamrx_amr_regrid(iamrx_amr a, int* changed)
{
    IAMRX_TRY
    const bool c = a->amr->regrid();
    if (c) amr_refresh_levels(a);
    if (changed) *changed = c ? 1 : 0;
    IAMRX_CATCH
}
int iamrx_amr_install_grids(iamrx_amr a, int nfine_levels, const int* nboxes, const int* boxes, int* changed)
{
    IAMRX_TRY
    std::vector<std::vector<BoxD>> g(nfine_levels);
    int q = 0;
    for (int l = 0; l < nfine_levels; ++l)
        for (int b = 0; b < nboxes[l]; ++b, ++q) {
            BoxD x;
            for (int d = 0; d < 3; ++d) { x.lo[d] = boxes[6 * q + d]; x.hi[d] = boxes[6 * q + 3 + d]; }
            g[l].push_back(x);
        }
    const uint64_t before = a->amr->grid_generation();
    bool c = false;
    try { c = a->amr->install_grids(g); }
    catch (...) { if (a->amr->grid_generation() != before) amr_refresh_levels(a); throw; }
    if (c) amr_refresh_levels(a);
    if (changed) *changed = c ? 1 : 0;
    IAMRX_CATCH
}
int iamrx_amr_nlevels(iamrx_amr a, int* nlev) { IAMRX_TRY *nlev = a->amr->nlevels(); IAMRX_CATCH }
int iamrx_amr_level_layout(iamrx_amr a, int lev, iamrx_layout* out)
{
    IAMRX_TRY
    auto* h = new iamrx_layout_s;
    h->p = a->amr->level(lev).user_layout;
    *out = h;
    IAMRX_CATCH
}
int iamrx_amr_level_boxes(iamrx_amr a, int lev, int* nboxes, int* boxes /* 6 ints per box, or NULL to query the count */)
{
    IAMRX_TRY
    const auto& bx = a->amr->level(lev).user_layout->boxes;
    if (boxes) {
        if (*nboxes < (int)bx.size()) throw Error("iamrx_amr_level_boxes: box capacity too small");
        for (size_t q = 0; q < bx.size(); ++q) for (int d = 0; d < 3; ++d) { boxes[6 * q + d] = bx[q].lo[d]; boxes[6 * q + 3 + d] = bx[q].hi[d]; }
    }
    *nboxes = (int)bx.size();
    IAMRX_CATCH
}
int iamrx_amr_level(iamrx_amr a, int lev, iamrx_ns* out) { IAMRX_TRY *out = a->levels.at(lev).get(); IAMRX_CATCH }
int iamrx_amr_post_init(iamrx_amr a, double stop_time) { IAMRX_TRY a->amr->post_init(stop_time); IAMRX_CATCH }
int iamrx_amr_coarse_step(iamrx_amr a, double* dt0)
{
    IAMRX_TRY
    const uint64_t before = a->amr->grid_generation();
    const double d = a->amr->coarse_step();
    if (a->amr->grid_generation() != before) amr_refresh_levels(a);      // the step regridded
    if (dt0) *dt0 = d;
    IAMRX_CATCH
}
int iamrx_amr_time(iamrx_amr a, double* time, double* dt_levels)
{
    IAMRX_TRY
    if (time) *time = a->amr->time();
    if (dt_levels) for (int l = 0; l < a->amr->nlevels(); ++l) dt_levels[l] = a->amr->dt(l);
    IAMRX_CATCH
}
int iamrx_amr_restart_state(iamrx_amr a, int set, double* dt_level, double* dt_min, int* n_cycle, int counters[2], double* stop_time)
{
    IAMRX_TRY
    if (set) a->amr->set_restart_state(dt_level, dt_min, n_cycle, counters, *stop_time);
    else a->amr->get_restart_state(dt_level, dt_min, n_cycle, counters, stop_time);
    IAMRX_CATCH
}
int iamrx_amr_level_counts(iamrx_amr a, int set, int* counts, int n)
{
    IAMRX_TRY
    if (set) a->amr->set_level_counts(counts, n); else a->amr->get_level_counts(counts, n);
    IAMRX_CATCH
}
int iamrx_ns_set_stop_time(iamrx_ns ns, double stop_time) { IAMRX_TRY ns->ns->set_stop_time(stop_time); IAMRX_CATCH }
int iamrx_amr_reflux(iamrx_amr a, int lev) { IAMRX_TRY a->amr->reflux(lev); IAMRX_CATCH }
int iamrx_amr_avg_down(iamrx_amr a, int lev) { IAMRX_TRY a->amr->avg_down(lev); IAMRX_CATCH }
int iamrx_amr_mac_sync(iamrx_amr a, int lev) { IAMRX_TRY a->amr->mac_sync(lev); IAMRX_CATCH }
int iamrx_amr_level_sync(iamrx_amr a, int lev) { IAMRX_TRY a->amr->level_sync(lev); IAMRX_CATCH }
int iamrx_amr_profile(iamrx_amr a, int enable, double sections_ms[16], double level_sections_ms /* [nlev][8] or NULL */[])
{
    IAMRX_TRY
    if (sections_ms) for (int i = 0; i < 16; ++i) sections_ms[i] = a->amr->t_prof[i];
    if (level_sections_ms) for (int l = 0; l < a->amr->nlevels(); ++l) for (int i = 0; i < 8; ++i) level_sections_ms[8 * l + i] = a->amr->level(l).t_sections[i];
    if (enable >= 0) a->amr->set_profile(enable != 0);
    IAMRX_CATCH
}
int iamrx_amr_sync_stats(iamrx_amr a, iamrx_mg_stats* sync, iamrx_mg_stats* mac_sync)
{
    IAMRX_TRY
    from_stats(a->amr->st_sync, sync); from_stats(a->amr->st_mac_sync, mac_sync);
    IAMRX_CATCH
}

int iamrx_ns_stats(iamrx_ns ns, iamrx_mg_stats* mac, iamrx_mg_stats* nodal, iamrx_mg_stats* visc)
{
    IAMRX_TRY
    from_stats(ns->ns->st_mac, mac); from_stats(ns->ns->st_nodal, nodal); from_stats(ns->ns->st_visc, visc);
    IAMRX_CATCH
}

int iamrx_kernel_probe_start(int which, long min_points, int stride)
{
    IAMRX_TRY
    if (which < 0 || which >= PROBE_COUNT) throw Error("iamrx_kernel_probe_start: unknown probe");
    kernel_probe_start(which, min_points, stride);
    IAMRX_CATCH
}
int iamrx_kernel_probe_stop(int which, double* total_ms, long* launches)
{
    IAMRX_TRY
    if (which < 0 || which >= PROBE_COUNT) throw Error("iamrx_kernel_probe_stop: unknown probe");
    kernel_probe_stop(which, total_ms, launches);
    IAMRX_CATCH
}

int iamrx_ns_profile(iamrx_ns ns, int enable, double sections_ms[8])
{
    IAMRX_TRY
    if (ns->probing) {          // close the kernel probe opened by enable = 3
        double ms = 0.0; long n = 0;
        gs4_probe_stop(&ms, &n);
        ns->ns->t_sections[6] = ms; ns->ns->t_sections[7] = (double)n;
        ns->probing = false;
    }
    if (sections_ms) for (int i = 0; i < 8; ++i) sections_ms[i] = ns->ns->t_sections[i];
    if (enable >= 0) ns->ns->profile_sections = enable == 1 || enable == 2;
    if (enable == 2) for (int i = 0; i < 8; ++i) ns->ns->t_sections[i] = 0.0;
    if (enable == 3) {
        const Layout& l = *ns->ns->lay();
        gs4_probe_start((long)(l.max_len[0] + 1) * (l.max_len[1] + 1) * (l.max_len[2] + 1), 8);
        ns->probing = true;
    }
    IAMRX_CATCH
}

}  // extern "C"
