// iamr_amd/csrc/nodalmg.hip -- nodal multigrid driver (MLMG on MLNodeLaplacian semantics) over the HIP
// kernels of k_nodal.hip.  Reference call site: Source/Projection.cpp:2512-2542 (NodalProjector::project
// with Gauss-Seidel on, harmonic average off, max_fmg_iter 0, proj_tol 1e-12 / sync_tol 1e-10).
#include "mlmg.h"
#include "launch.h"
#include "krylov.h"
#include <chrono>
#include <cmath>
#include <cstdlib>

namespace iamrx {

// kernels (k_nodal.hip)
void nodal_restrict(MultiFab& crse, const MultiFab& fine);
void nodal_interp_add(MultiFab& fine, const MultiFab& crse, const MultiFab& sig_fine);

static bool nodal_fused()
{
    static int v = -1;
    // plane-fused sweep (2 launches + 2 fills instead of 8 + 8): measured per sweep on MI355X 0.69 vs 0.79 ms at 256^3,
    // 0.11 vs 0.20 ms at 128^3, 0.03 vs 0.09 ms at <= 64^3.  IAMRX_NODAL_FUSED=0 selects the 8 colour passes.
    v = tune("NODAL_FUSED", 1) != 0 ? 1 : 0;
    return v == 1;
}
static bool nodal_small()
{
    static int v = -1;
    v = tune("NODAL_SMALL", 1) != 0 ? 1 : 0;
    return v == 1;
}
bool nodal_smooth_small(const Geometry& g, MultiFab& x, const MultiFab& rhs, const MultiFab& sig, int nsweeps);
bool nodal_bottom_device_ok(const Geometry& g, const Layout& l);
void nodal_bottom_solve(const Geometry& g, MultiFab& cor, const MultiFab& res, const MultiFab& sig, bool singular, double eps_rel, int maxiter,
                        int nsweeps, int nub, int nuf, int* d_iters);
bool nodal_bottom_device_ok_general(const Geometry& g, const Layout& l);
void nodal_bottom_solve_general(const Geometry& g, MultiFab& cor, const MultiFab& res, const MultiFab& sig, const MultiFab* dmask, bool singular,
                                double eps_rel, int maxiter, int nsweeps, int nub, int nuf, int* d_iters);
static int* nodal_bottom_iters_dev()
{
    static int* d = nullptr;
    if (!d) { IAMRX_HIP_CHECK(hipMalloc(&d, sizeof(int))); IAMRX_HIP_CHECK(hipMemset(d, 0, sizeof(int))); }
    return d;
}

NodalMG::NodalMG(const Geometry& g, LayoutP layout, const DomainBC& bc_in, const MGOpts& o) : m_g(g), m_bc(bc_in), m_o(o)
{
    ProfScope ps_prof_("nmg_ctor");
    // the operator does not distinguish inflow faces from walls (the difference is in div(u), nodal_divu)
    for (int d = 0; d < 3; ++d) { if (m_bc.lo[d] == lo_inflow) m_bc.lo[d] = lo_neumann; if (m_bc.hi[d] == lo_inflow) m_bc.hi[d] = lo_neumann; }
    const DomainBC& bc = m_bc;
    for (int d = 0; d < 3; ++d) {
        if (!g.periodic[d])
            for (int t : {bc.lo[d], bc.hi[d]})
                if (t != lo_neumann && t != lo_dirichlet) throw Error("iamrx NodalMG: domain boundaries must be periodic, Neumann (wall / inflow) or Dirichlet (outflow)");
        // Neumann-wall nodes carry weight 1/2 in sums and dot products (doubled rows, MLNodeLinOp dot mask)
        m_g.half_lo[d] = (!g.periodic[d] && bc.lo[d] == lo_neumann) ? 1 : 0;
        m_g.half_hi[d] = (!g.periodic[d] && bc.hi[d] == lo_neumann) ? 1 : 0;
    }
    m_lev.resize(1);
    m_lev[0].g = m_g;
    m_lev[0].layout = std::move(layout);
    while ((int)m_lev.size() <= m_o.max_coarsening_level) {
        Level& f = m_lev.back();
        // a fully periodic single box of at most 8^3 cells is solved by the single-workgroup device bottom solver (k_nodal_bottom)
        if (m_o.device_bottom && m_o.nodal_smoother == 0 && !m_o.bottom_smoother_only &&
            (nodal_bottom_device_ok(f.g, *f.layout) || nodal_bottom_device_ok_general(f.g, *f.layout))) break;
        bool dom_ok = true;
        for (int d = 0; d < 3; ++d) if (f.g.domain.len(d) % 2 != 0 || f.g.domain.len(d) / 2 < m_o.min_width) dom_ok = false;
        const bool iso = dom_ok && f.layout->coarsenable(2, m_o.min_width);
        const bool slab = !iso && mg_slab_level(f.g, *f.layout, m_o.min_width, m_o.slab != 0);     // (mlmg.hip: y kept at two cells, transfers through the one-plane level)
        if (!iso && !slab) break;
        Level c;
        c.g = f.g;
        if (slab) {
            const Geometry sg = mg_slab_geom(f.g);
            c.g.domain = sg.domain;
            for (int d = 0; d < 3; ++d) c.g.dx[d] = sg.dx[d];
            c.slab = true;
            c.virt = f.layout->coarsened(2);
            c.layout = f.layout->slab_coarsened();
        } else {
        c.g.domain = coarsen(f.g.domain, 2);
        for (int d = 0; d < 3; ++d) c.g.dx[d] = f.g.dx[d] * 2.0;
        c.layout = f.layout->coarsened(2);
        }
        if (mg_agglomerate_level(*c.layout)) {
            c.agg = true;
            c.dist = c.layout;
            c.layout = c.dist->make_replicated();
        }
        m_lev.push_back(std::move(c));
    }
    for (auto& L : m_lev) {
        if (L.agg) L.tmp_d.define(L.dist, node_type(), 1, 1);
        if (L.slab) L.vres.define(L.virt, node_type(), 1, 0);
        // 4 ghost layers: the plane-fused Gauss-Seidel recomputes its halo instead of exchanging it per colour
        const int ng = nodal_fused() ? 4 : 1;
        L.sig.define(L.layout, cell_type(), 1, ng);
        // ghost cells beyond a coarse/fine boundary are never filled (setSigma: valid cells, neighbours / periodic images, wall mirrors);
        // the prolongation forms sigma-weighted averages at the (masked) boundary nodes before they are zeroed: keep them finite
        L.sig.setVal(0.0);
        L.cor.define(L.layout, node_type(), 1, ng);
        L.res.define(L.layout, node_type(), 1, ng);
        L.rescor.define(L.layout, node_type(), 1, 1);
        L.cor.setVal(0.0); L.res.setVal(0.0); L.rescor.setVal(0.0);
    }
    // Dirichlet nodes: on Dirichlet (outflow) domain faces and on the boundary of a level that does not cover the domain
    // (coarse/fine boundary of an AMR level).  They keep their value, carry no residual and take no correction.
    bool need_mask = false;
    for (int d = 0; d < 3; ++d) if (!g.periodic[d] && (bc.lo[d] == lo_dirichlet || bc.hi[d] == lo_dirichlet)) need_mask = true;
    if (m_lev[0].layout->total_cells() != g.domain.npts()) need_mask = true;
    for (auto& L : m_lev) {
        if (!need_mask) break;
        MultiFab cov(L.layout, cell_type(), 1, 1);
        cov.setVal(0.0);
        mf_add_scalar(cov, 1.0, 0, 1, 0);
        cov.FillBoundary(L.g);
        const int ng = L.cor.ngrow;
        L.dm.define(L.layout, node_type(), 1, ng);
        L.dm.setVal(1.0);                                  // ghost nodes outside the level: never updated
        nodal_build_dmask(L.g, L.dm, cov, m_bc);
        if (L.dm.norm0(0, 1, 0) == 0.0) L.dm = MultiFab();
        else {
            m_masked = true;
            L.dm.FillBoundary(L.g);
            nodal_reflect_bc(L.g, L.dm, m_bc);
        }
    }
    if (m_masked) {
        m_singular = false;
        for (auto& L : m_lev) IAMRX_ASSERT(L.dm.defined());
    }
}

void NodalMG::setSigma(const MultiFab& sig, int comp)
{
    ProfScope ps_prof_("nmg_setsigma");
    MultiFab::Copy(m_lev[0].sig, sig, comp, 0, 1, 0);
    m_lev[0].sig.FillBoundary(m_lev[0].g);
    cc_mirror_bc(m_lev[0].g, m_lev[0].sig);                // mlndlap_fillbc_cc: mirror sigma across walls
    for (size_t l = 1; l < m_lev.size(); ++l) {
        // arithmetic average (harmonic averaging off); agglomerated level: on its distributed form, then gathered; slab level: onto the
        // one-plane virtual level, duplicated
        {
            Level& C = m_lev[l];
            MultiFab sd, sv;
            MultiFab* held = &C.sig;
            if (C.agg) { sd.define(C.dist, cell_type(), 1, 0); held = &sd; }
            if (C.slab) { sv.define(C.virt, cell_type(), 1, 0); cc_restrict(sv, m_lev[l - 1].sig); slab_duplicate(*held, sv); }
            else cc_restrict(*held, m_lev[l - 1].sig);
            if (C.agg) gather_to_replicated(C.sig, sd);
        }
        m_lev[l].sig.FillBoundary(m_lev[l].g);
        cc_mirror_bc(m_lev[l].g, m_lev[l].sig);
    }
    // Constant sigma (constant-density flow, the common IAMR case): every coarsened level then holds the same constant (the average
    // of 8 equal numbers is exact) and periodic / mirrored ghost cells too, so the smoother can take sigma from a register instead
    // of staging two sigma planes per node plane.  Same expression tree => bit-identical results.  IAMRX_NODAL_CSIG=0 disables.
    const bool csig_on = tune("NODAL_CSIG", 1) != 0;
    m_csig = false;
    if (csig_on && !m_masked) {
        // (largest and smallest value in one pass and one read-back: sigma changes with every projection, the test runs in front of each)
        double smin, smax;
        reduce_minmax(m_lev[0].sig, 0, 0, smin, smax);
        if (smax > 0.0 && smin == smax) { m_csig = true; m_csig_val = smax; }
    }
}

// ghost nodes: same-level + periodic images, then even reflection about Neumann walls
void NodalMG::fillbc(int l, MultiFab& x, int kpar, hipStream_t on)
{
    // the plane-fused smoother recomputes a 4-node halo in-plane but reaches only one plane up and down: exchange 1 plane in z
    const int ngv[3] = {x.ngrow, x.ngrow, 1};
    x.FillBoundary(m_lev[l].g, 0, x.ncomp, ngv, kpar, on);
    nodal_reflect_bc(m_lev[l].g, x, m_bc, on);
}

void NodalMG::smooth(int l, MultiFab& x, const MultiFab& rhs, bool x_is_zero, bool leave_ghosts)
{
    Level& L = m_lev[l];
    // a correction that starts from zero: on a level the register-resident kernel smooths with index wrap (no ghost nodes are read) the
    // first sweep is told so and reads no x -- the zero fill and a third of the first sweep's traffic; everywhere else x is zeroed here
    // (with index wrap or on ghost-filled boxes, with or without a Dirichlet mask: the ghost nodes of a zero array are zero as well --
    // images, reflections at walls -- so the fill in front of the first pass goes too; IAMRX_NODAL_ZERO_START = 2: index-wrap levels only)
    const int zs_mode = (int)tune("NODAL_ZERO_START", 1);
    int zs_refl = 0;
    const bool zero_start = x_is_zero && m_o.nodal_smoother == 0 && nodal_fused() && zs_mode != 0 && nodal_gsr_applies(x, rhs, L.dmask()) &&
                            (zs_mode != 2 || (!L.dmask() && nodal_wrap_or_reflect_ok(L.g, *L.layout, m_bc, 4, &zs_refl)));       // (gsr_applies: boxes >= 48 cells -- never the single-workgroup smoother's level)
    if (x_is_zero && !zero_start) x.setVal(0.0);
    // small single-box periodic levels: all sweeps x colours in one single-workgroup launch
    const MultiFab* dmk = L.dmask();
    if (!dmk && m_o.nodal_smoother == 0 && nodal_small() && nodal_smooth_small(L.g, x, rhs, L.sig, m_o.nodal_sweeps)) {
        fillbc(l, x);
        return;
    }
    if (m_o.nodal_smoother == 0 && nodal_fused()) {
        // colours 0-3 (k even) in one pass, colours 4-7 (k odd) in a second one: identical arithmetic to the eight
        // sequential colour passes below.  Each sweep goes from one buffer to the other (see k_nodal_gs4).
        if (!L.xb.defined() || L.xb.ngrow != x.ngrow) L.xb.define(L.layout, node_type(), 1, x.ngrow);
        // one box spanning a fully periodic domain: the kernel takes periodic images from the valid data, no ghost fills
        // ... or a domain whose non-periodic directions end on Neumann walls: mirror images in those directions (k_nodal.hip image_node)
        int refl = 0;
        const bool wrap = !dmk && nodal_wrap_or_reflect_ok(L.g, *L.layout, m_bc, 4, &refl);
        if (!wrap) {
            // the right-hand side of the level's smooth calls is its residual array, unchanged within a V-cycle: fill its ghosts once
            if (&rhs == &L.res) { if (!L.res_filled) { fillbc(l, L.res); L.res_filled = true; } }
            else fillbc(l, const_cast<MultiFab&>(rhs));
        }
        MultiFab* a = &x;
        MultiFab* b = &L.xb;
        // Ghost traffic: the even pass changes even planes only and reads the odd planes next to them, the odd pass the reverse.
        // After the first (full) fill it is therefore enough to refresh the ghost nodes of the planes of ONE parity in front of each
        // pass: the odd planes before the even pass, the even planes before the odd pass (half the halo volume; on boxes stacked in
        // z, where one ghost plane is exchanged, every second message disappears).
        const bool par_fill = tune("NODAL_PARITY_FILL", 1) != 0;
        // Overlap (IAMRX_HALO_OVERLAP as for the cell-centred sweep, CellMG::smooth_n; round 6): a pass of k_nodal_gsr whose ghost nodes have
        // to be refreshed first is issued in two parts -- the tiles whose footprint and planes lie inside their box on the main stream; the
        // exchange, the wall reflection and the remaining tiles on the context's side stream behind a fork -- and joined: the messages
        // travel while the interior tiles run.  The two parts write disjoint nodes; the same doubles as the one-piece pass.
        auto& ctx = Context::get();
        const int ov_mode = (int)tune("HALO_OVERLAP", 1);
        bool overlap = false;
        if (!wrap && ov_mode != 0 && nodal_gsr_splits(x, rhs, dmk)) {
            const int ngv[3] = {x.ngrow, x.ngrow, 1};
            // (the plans are built and uploaded in front of any fork)
            bool peers = false;
            for (int kp = -1; kp <= 1; ++kp) peers = !fill_boundary_plan(*L.layout, node_type(), x.ngrow, L.g, ngv, kp).peers.empty() || peers;
            overlap = ov_mode == 2 || peers;
        }
        auto pass = [&](MultiFab& filled, int fill_kpar, bool do_fill, const MultiFab& xc_, const MultiFab& xn_, MultiFab& xo_, int kpar, int zf) {
            const double* cs = m_csig ? &m_csig_val : nullptr;
            if (do_fill && overlap) {
                ctx.fork_side();
                nodal_gs_fused_pass(L.g, xc_, xn_, xo_, rhs, L.sig, kpar, wrap, dmk, cs, zf, refl, 1, ctx.stream);
                fillbc(l, filled, fill_kpar, ctx.side);
                nodal_gs_fused_pass(L.g, xc_, xn_, xo_, rhs, L.sig, kpar, wrap, dmk, cs, zf, refl, 2, ctx.side);
                ctx.join_side();
                return;
            }
            if (do_fill) fillbc(l, filled, fill_kpar);
            nodal_gs_fused_pass(L.g, xc_, xn_, xo_, rhs, L.sig, kpar, wrap, dmk, cs, zf, refl);
        };
        for (int ns = 0; ns < m_o.nodal_sweeps; ++ns) {
            const bool z = zero_start && ns == 0;
            // even planes: a -> b (ghost nodes of a refreshed first: all planes in front of the first sweep, then the odd ones)
            pass(*a, (ns == 0 || !par_fill) ? -1 : 1, !wrap && !z, *a, *a, *b, 0, z ? 3 : 0);
            // odd planes: centre from a, neighbours from b (the ghost images of b's new even planes first)
            pass(*b, par_fill ? 0 : -1, !wrap, *a, *b, *b, 1, z ? 1 : 0);
            std::swap(a, b);
        }
        if (a != &x) MultiFab::Copy(x, *a, 0, 0, 1, 0);
        if (!leave_ghosts) fillbc(l, x);
        return;
    }
    for (int ns = 0; ns < m_o.nodal_sweeps; ++ns) {
        if (false) {
        } else if (m_o.nodal_smoother == 0) {
            for (int color = 0; color < 8; ++color) {
                fillbc(l, x);
                nodal_gs_color(L.g, x, rhs, L.sig, color, dmk);
            }
        } else {
            if (!L.tmp.defined()) L.tmp.define(L.layout, node_type(), 1, 1);
            fillbc(l, x);
            nodal_jacobi(L.g, L.tmp, x, rhs, L.sig, dmk);
            MultiFab::Copy(x, L.tmp, 0, 0, 1, 0);
        }
    }
    fillbc(l, x);
}

void NodalMG::residual(int l, MultiFab& r, MultiFab& x, const MultiFab& b, double* norm, bool x_filled)
{
    if (!x_filled) fillbc(l, x);
    const bool masked = (bool)m_lev[l].dmask();
    const bool have = nodal_residual(m_lev[l].g, r, x, m_lev[l].sig, &b, (norm && !masked) ? norm : nullptr);
    if (masked) nodal_zero_masked(r, m_lev[l].dm);
    if (norm && !have) *norm = r.norm0(0, 1, 0);
}

void NodalMG::subtract_mean(int l, MultiFab& mf)
{
    const Geometry& g = m_lev[l].g;
    double cnt = 1.0;
    for (int d = 0; d < 3; ++d) cnt *= (double)g.domain.len(d);   // sum of weights: n unique nodes (periodic) or (n-1) + 2*(1/2) (walls)
    const double s = mf.sum_unique(g, 0);
    mf_add_scalar(mf, -s / cnt, 0, 1, 0);
}

int NodalMG::bicgstab(int l, MultiFab& sol, const MultiFab& rhs, double eps_rel, double eps_abs, int& niters)
{
    Level& L = m_lev[l];
    const Geometry& g = L.g;
    auto mk = [&](int ng) { return MultiFab(L.layout, node_type(), 1, ng); };
    MultiFab ph = mk(1), sh = mk(1), sorig = mk(0), p = mk(0), r = mk(0), s = mk(0), rh = mk(0), v = mk(0), t = mk(0);
    ph.setVal(0.0); sh.setVal(0.0);
    residual(l, r, sol, rhs);
    MultiFab::Copy(sorig, sol, 0, 0, 1, 0);
    MultiFab::Copy(rh, r, 0, 0, 1, 0);
    sol.setVal(0.0);
    double rnorm = r.norm0(0, 1, 0);
    const double rnorm0 = rnorm;
    int ret = 0, nit = 1;
    double rho_1 = 0, alpha = 0, omega = 0;
    if (rnorm0 == 0 || rnorm0 < eps_abs) { niters = 0; MultiFab::Copy(sol, sorig, 0, 0, 1, 0); return 0; }
    // Krylov bound (see CellMG::bicgstab): cap at twice the number of unique nodes of the bottom level
    long nunk = 1;
    for (int d = 0; d < 3; ++d) nunk *= g.domain.len(d) + (g.periodic[d] ? 0 : 1);
    const int maxiter = (int)std::min<long>(m_o.bottom_maxiter, std::max<long>(8, 2 * nunk));
    if (tune("KRYLOV_DEVICE", 1) != 0 && (Context::get().comm->nranks == 1 || L.layout->replicated)) {
        // krylov.h: the same loop with its scalars on the device (one status word per iteration comes back, one iteration late)
        ret = bicgstab_device(*L.layout, node_type(), 1, g, sol, r, rh, ph, sh, v, t, rnorm0, eps_rel, eps_abs, maxiter,
                              [&](MultiFab& out, MultiFab& in) {
                                  fillbc(l, in);
                                  nodal_residual(g, out, in, L.sig, nullptr);
                                  if (L.dmask()) nodal_zero_masked(out, L.dm);
                              }, nit, rnorm);
    } else
    for (; nit <= maxiter; ++nit) {
        double rho;
        { const MultiFab* xs[1] = {&rh}; const MultiFab* ys[1] = {&r}; reduce_dots(1, xs, ys, 0, 1, g, &rho); }
        if (rho == 0) { ret = 1; break; }
        if (nit == 1) MultiFab::Copy(p, r, 0, 0, 1, 0);
        else {
            const double beta = (rho / rho_1) * (alpha / omega);
            mf_lincomb(p, 1.0, p, -omega, v, 0, 1, 0);
            mf_lincomb(p, 1.0, r, beta, p, 0, 1, 0);
        }
        MultiFab::Copy(ph, p, 0, 0, 1, 0);
        fillbc(l, ph);
        nodal_residual(g, v, ph, L.sig, nullptr);
        if (L.dmask()) nodal_zero_masked(v, L.dm);
        double rhTv;
        { const MultiFab* xs[1] = {&rh}; const MultiFab* ys[1] = {&v}; reduce_dots(1, xs, ys, 0, 1, g, &rhTv); }
        if (rhTv != 0) alpha = rho / rhTv; else { ret = 2; break; }
        mf_lincomb(sol, 1.0, sol, alpha, ph, 0, 1, 0);
        mf_lincomb(s, 1.0, r, -alpha, v, 0, 1, 0);
        rnorm = s.norm0(0, 1, 0);
        if (rnorm < eps_rel * rnorm0 || rnorm < eps_abs) break;
        MultiFab::Copy(sh, s, 0, 0, 1, 0);
        fillbc(l, sh);
        nodal_residual(g, t, sh, L.sig, nullptr);
        if (L.dmask()) nodal_zero_masked(t, L.dm);
        double tv[2];
        { const MultiFab* xs[2] = {&t, &t}; const MultiFab* ys[2] = {&t, &s}; reduce_dots(2, xs, ys, 0, 1, g, tv); }
        if (tv[0] != 0) omega = tv[1] / tv[0]; else { ret = 3; break; }
        mf_lincomb(sol, 1.0, sol, omega, sh, 0, 1, 0);
        mf_lincomb(r, 1.0, s, -omega, t, 0, 1, 0);
        rnorm = r.norm0(0, 1, 0);
        if (rnorm < eps_rel * rnorm0 || rnorm < eps_abs) break;
        if (omega == 0) { ret = 4; break; }
        rho_1 = rho;
    }
    if (ret == 0 && rnorm > eps_rel * rnorm0 && rnorm > eps_abs) ret = 8;
    if ((ret == 0 || ret == 8) && rnorm < rnorm0) mf_lincomb(sol, 1.0, sol, 1.0, sorig, 0, 1, 0);
    else { sol.setVal(0.0); mf_lincomb(sol, 1.0, sol, 1.0, sorig, 0, 1, 0); }
    niters = nit;
    return ret;
}

bool NodalMG::bottom_on_device()
{
    Level& B = m_lev.back();
    if (!(m_o.device_bottom && m_o.nodal_smoother == 0 && !m_o.bottom_smoother_only)) return false;
    return (!B.dmask() && nodal_bottom_device_ok(B.g, *B.layout)) || nodal_bottom_device_ok_general(B.g, *B.layout);
}

void NodalMG::vcycle(MGStats& st)
{
    const int nl = (int)m_lev.size();
    for (int l = 0; l < nl - 1; ++l) {
        Level& L = m_lev[l];
        if (m_o.nodal_nu1 <= 0) L.cor.setVal(0.0);
        for (int i = 0; i < m_o.nodal_nu1; ++i) smooth(l, L.cor, L.res, i == 0);
        residual(l, L.rescor, L.cor, L.res, nullptr, m_o.nodal_nu1 > 0);          // (smooth() has just filled the ghost nodes)
        fillbc(l, L.rescor);
        {
            Level& C = m_lev[l + 1];
            MultiFab& held = C.agg ? C.tmp_d : C.res;
            nodal_restrict(C.slab ? C.vres : held, L.rescor);
            if (C.slab) slab_duplicate(held, C.vres);
            if (C.agg) gather_to_replicated(C.res, C.tmp_d);
        }
        m_lev[l + 1].res_filled = false;
        if (m_lev[l + 1].dmask()) nodal_zero_masked(m_lev[l + 1].res, m_lev[l + 1].dm);   // mlndlap_restriction: 0 on Dirichlet nodes
    }
    {
        const int l = nl - 1;
        Level& B = m_lev[l];
        B.cor.setVal(0.0);
        if (m_o.bottom_smoother_only) {
            for (int i = 0; i < m_o.nuf; ++i) smooth(l, B.cor, B.res);
        } else if (bottom_on_device()) {
            long nunk = 1;
            for (int d = 0; d < 3; ++d) nunk *= B.g.domain.len(d) + (B.g.periodic[d] ? 0 : 1);
            const int maxiter = (int)std::min<long>(m_o.bottom_maxiter, std::max<long>(8, 2 * nunk));
            if (!B.dmask() && nodal_bottom_device_ok(B.g, *B.layout))           // fully periodic: the wrap-only kernel
                nodal_bottom_solve(B.g, B.cor, B.res, B.sig, m_singular, m_o.bottom_reltol, maxiter, m_o.nodal_sweeps, m_o.nub, m_o.nuf,
                                   nodal_bottom_iters_dev());
            else                                                                // walls / Dirichlet mask / refined patch
                nodal_bottom_solve_general(B.g, B.cor, B.res, B.sig, B.dmask(), m_singular, m_o.bottom_reltol, maxiter, m_o.nodal_sweeps, m_o.nub,
                                           m_o.nuf, nodal_bottom_iters_dev());
        } else {
            MultiFab rb(B.layout, node_type(), 1, 0);
            MultiFab::Copy(rb, B.res, 0, 0, 1, 0);
            if (m_singular) subtract_mean(l, rb);
            int nit = 0;
            const int ret = bicgstab(l, B.cor, rb, m_o.bottom_reltol, -1.0, nit);
            st.bottom_iters_total += nit;
            if (ret != 0) {
                B.cor.setVal(0.0);
                for (int i = 0; i < m_o.nuf; ++i) smooth(l, B.cor, B.res);
            }
            const int nn = ret == 0 ? m_o.nub : m_o.nuf;
            for (int i = 0; i < nn; ++i) smooth(l, B.cor, B.res);
        }
    }
    for (int l = nl - 2; l >= 0; --l) {
        Level& L = m_lev[l];
        // (a level that was smoothed on the way up comes with its ghost nodes filled; the bottom level comes from its solver)
        if (l + 1 == nl - 1 || m_o.nodal_nu2 <= 0) fillbc(l + 1, m_lev[l + 1].cor);
        if (m_lev[l + 1].agg) {
            scatter_from_replicated(m_lev[l + 1].tmp_d, m_lev[l + 1].cor, 1);
            nodal_interp_add(L.cor, m_lev[l + 1].tmp_d, L.sig);
        } else
        nodal_interp_add(L.cor, m_lev[l + 1].cor, L.sig);
        if (L.dmask()) nodal_zero_masked(L.cor, L.dm);                  // mlndlap_interpadd: Dirichlet nodes take no correction
        // the finest level's correction is added to the solution node by node: nobody reads its ghost nodes
        for (int i = 0; i < m_o.nodal_nu2; ++i) smooth(l, L.cor, L.res, false, l == 0 && i == m_o.nodal_nu2 - 1);
    }
}

void NodalMG::vcycle_correction(MultiFab& e, const MultiFab& r, MGStats& st)
{
    Level& L0 = m_lev[0];
    MultiFab::Copy(L0.res, r, 0, 0, 1, 0);
    if (L0.dmask()) nodal_zero_masked(L0.res, L0.dm);
    if (m_singular) subtract_mean(0, L0.res);
    L0.res_filled = false;
    vcycle(st);
    MultiFab::Copy(e, L0.cor, 0, 0, 1, 0);
    e.FillBoundary(L0.g);
    nodal_reflect_bc(L0.g, e, m_bc);
}

// the same on the solver's own arrays: the caller has written r to the valid nodes of res(0) and reads e from cor(0) (valid nodes + one
// layer of ghost nodes) -- no copy in, no copy out (the composite solver calls this once per level correction)
void NodalMG::vcycle_correction_inplace(MGStats& st)
{
    Level& L0 = m_lev[0];
    if (L0.dmask()) nodal_zero_masked(L0.res, L0.dm);
    if (m_singular) subtract_mean(0, L0.res);
    L0.res_filled = false;
    vcycle(st);
    const int ngv[3] = {1, 1, 1};
    L0.cor.FillBoundary(L0.g, 0, 1, ngv, -1);
    nodal_reflect_bc(L0.g, L0.cor, m_bc);
}

MGStats NodalMG::solve(MultiFab& phi, const MultiFab& rhs_in, double rtol, double atol)
{
    ProfScope ps_prof_("nmg_solve");
    auto& ctx = Context::get();
    MGStats st;
    st.nlevels = (int)m_lev.size();
    Level& L0 = m_lev[0];
    MultiFab rhs(L0.layout, node_type(), 1, 0);
    MultiFab::Copy(rhs, rhs_in, 0, 0, 1, 0);
    if (L0.dmask()) nodal_zero_masked(rhs, L0.dm);
    if (m_singular) subtract_mean(0, rhs);
    residual(0, L0.res, phi, rhs, &st.resnorm0);
    L0.res_filled = false;
    st.rhsnorm0 = rhs.norm0(0, 1, 0);
    const double max_norm = st.rhsnorm0 >= st.resnorm0 ? st.rhsnorm0 : st.resnorm0;
    const double res_target = std::max(atol, std::max(rtol, 1.e-16) * max_norm);
    st.resnorm = st.resnorm0;
    if (m_o.verbose) printf("iamrx nodal MLMG: rhs %.6e resid0 %.6e levels %d (fused sweep %d, single-workgroup coarse smoother %d, ghost width %d)\n",
                            st.rhsnorm0, st.resnorm0, st.nlevels, (int)nodal_fused(), (int)nodal_small(), L0.cor.ngrow);
    double vc_ms = 0.0;
    cycle_timer().used = 0;
    const bool bdev = bottom_on_device();
    if (bdev) IAMRX_HIP_CHECK(hipMemsetAsync(nodal_bottom_iters_dev(), 0, sizeof(int), ctx.stream));
    if (m_o.fixed_iters <= 0 && st.resnorm0 <= res_target) st.converged = 1;
    else {
        const int maxit = m_o.fixed_iters > 0 ? m_o.fixed_iters : m_o.max_iters;
        std::vector<double> hist;
        for (int iter = 0; iter < maxit; ++iter) {
            // see CellMG::solve: the mean of the residual of a singular system is removed in front of the first cycle only
            const bool all_per = L0.g.periodic[0] && L0.g.periodic[1] && L0.g.periodic[2] && !L0.dmask();
            // (levels with Neumann walls keep amrex::MLMG's per-iteration removal: ADVICE round 3)
            if (m_singular && (!all_per || tune("MG_RES_MEAN", 0) != 0)) { subtract_mean(0, L0.res); L0.res_filled = false; }
            cycle_timer().mark(ctx.stream);
            vcycle(st);
            cycle_timer().mark(ctx.stream);
            mf_saxpy(phi, 1.0, L0.cor, 0, 0, 1, 0);
            residual(0, L0.res, phi, rhs, &st.resnorm);
            L0.res_filled = false;
            st.iters = iter + 1;
            if (m_o.verbose) printf("iamrx nodal MLMG: iter %d resid %.6e\n", iter + 1, st.resnorm);
            if (m_o.fixed_iters <= 0 && st.resnorm <= res_target) { st.converged = 1; break; }
            if (!(st.resnorm < 1.e20 * max_norm)) throw Error("iamrx nodal MLMG: failing to converge (residual blow-up)");
            // stalled at the fp64 round-off floor just above the tolerance: see composite_project (amrns.hip); converged = 2
            hist.push_back(st.resnorm);
            if (m_o.fixed_iters <= 0 && mg_stalled_at_floor(hist, st.resnorm0, res_target, tune("MG_STALL_FAST", 1) != 0)) {
                st.converged = 2;
                fprintf(stderr, "iamrx nodal MLMG: WARNING: residual %.3e stalled at the round-off floor above the target %.3e after %d cycles; accepted (converged = 2)\n",
                        st.resnorm, res_target, st.iters);
                break;
            }
        }
        if (m_o.fixed_iters <= 0 && !st.converged) throw Error("iamrx nodal MLMG: failed to converge after max_iters");
    }
    vc_ms = cycle_timer().total_ms();          // the last residual norm has synchronised the stream
    if (st.iters > 0) st.vcycle_ms = vc_ms / st.iters;
    if (bdev && st.iters > 0) {
        int h = 0;
        IAMRX_HIP_CHECK(hipMemcpyAsync(&h, nodal_bottom_iters_dev(), sizeof(int), hipMemcpyDeviceToHost, ctx.stream));
        ctx.sync();
        st.bottom_iters_total = h;
    }
    fillbc(0, phi);
    return st;
}

}  // namespace iamrx
