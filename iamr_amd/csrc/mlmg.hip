// iamr_amd/csrc/mlmg.hip -- cell-centred multigrid driver: MLMG::solve / oneIter / mgVcycle /
// actualBottomSolve and MLCGSolver::solve_bicgstab semantics (upstream AMReX), driving the HIP kernels
// of k_abec.hip / k_tensor.hip.  Reference call sites: Source/MacProj.cpp:1150-1183 (MAC solve,
// mac_tol 1e-12, max_order 4), Source/Diffusion.cpp:837-929 (tensor solve, visc_tol 1e-10, max_order 2).
#include "mlmg.h"
#include "krylov.h"
#include "launch.h"
#include <chrono>
#include <cmath>

namespace iamrx {

// amr.hip
void parallel_copy(MultiFab& dst, const MultiFab& src, int scomp, int dcomp, int nc, int src_ng, int dst_ng, const Geometry* periodic_geom, bool add);

CycleTimer& cycle_timer() { static CycleTimer t; return t; }

long mg_agglomeration_cells()
{
    static long v = -1;
    v = (long)tune("MG_AGGLOMERATE_CELLS", 2097152.0);
    return v;
}

// A coarse multigrid level is agglomerated -- continued on a layout that holds the whole level on this rank, its boxes merged
// (Layout::make_replicated) -- when it is small (IAMRX_MG_AGGLOMERATE_CELLS) and either spread over several ranks or, on one rank, made of
// several boxes (IAMRX_MG_AGG_SINGLE_RANK, 1): below that level no ghost exchange between boxes, one box per level that covers its domain
bool mg_agglomerate_level(const Layout& c)
{
    if (c.replicated || c.total_cells() > mg_agglomeration_cells()) return false;
    if (Context::get().comm->nranks > 1) return true;
    return c.boxes.size() > 1 && tune("MG_AGG_SINGLE_RANK", 1) != 0;
}

// Slab levels (IAMRX_MG_SLAB, set by the driver for 2-D inputs lifted onto a y-periodic slab): the level is two
// cells thick in y -- periodic, every box spanning it -- and cannot be coarsened there any more, but it can in x and z.  The multigrid then
// continues with y KEPT at two cells and dx doubled in every direction: for fields that do not vary along the slab the y-terms of every
// operator vanish, the two-plane level is the 2-D coarse problem (relaxed like the thick slab is), and the transfers are the ordinary ones
// onto the one-plane coarsening of the level with its plane duplicated (Layout::slab_coarsened, slab_duplicate) -- which also projects
// any y-dependent round-off out of the coarse levels, where the doubled dy would over-correct it.
bool mg_slab_level(const Geometry& g, const Layout& l, int min_width, bool slab_problem)
{
    if (!slab_problem && tune("MG_SLAB", 0) == 0) return false;
    if (!g.periodic[1] || g.domain.lo[1] != 0 || g.domain.hi[1] != 1) return false;
    for (int d = 0; d < 3; d += 2) if (g.domain.len(d) % 2 != 0 || g.domain.len(d) / 2 < min_width) return false;
    return l.slab_coarsenable(min_width);
}
Geometry mg_slab_geom(const Geometry& f)
{
    Geometry c = f;
    c.domain = coarsen(f.domain, 2);
    c.domain.lo[1] = 0; c.domain.hi[1] = 1;
    for (int d = 0; d < 3; ++d) c.dx[d] = f.dx[d] * 2.0;
    return c;
}

CellMG::CellMG(const Geometry& g, LayoutP layout, int ncomp, const DomainBC& bc, const MGOpts& o)
    : m_g(g), m_ncomp(ncomp), m_o(o)
{
    m_bcn.assign(1, bc);
    m_lev.resize(1);
    m_lev[0].g = g;
    m_lev[0].layout = std::move(layout);
}

// residual / apply of a level: the constant-viscosity tensor operator in one fused launch, everything else through abec_residual
static void level_residual(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& phi, const MultiFab* rhs, double* norm_out = nullptr)
{
    if (c.tensor && tensor_residual_fused(g, c, out, phi, rhs, norm_out)) return;
    abec_residual(g, c, out, phi, rhs, norm_out);
}

AbecCoef CellMG::coef(int l) const
{
    AbecCoef c;
    c.alpha = m_alpha; c.beta = m_beta; c.tensor = m_tensor ? 1 : 0;
    if (l == 0) {
        c.a = (m_nbr && m_a1.defined()) ? &m_a1 : m_a0;
        for (int d = 0; d < 3; ++d) c.b[d] = m_b0[d];
        c.tensor_eta = m_tensor_eta ? 1 : 0;
        c.sig = m_sig; c.sig_comp = m_sig_comp; c.sig_scale = m_sig_scale;
        if (m_nbr && m_sig) { c.sig = &m_sig2; c.sig_comp = 0; }
        c.b_uniform = m_buni ? 1 : 0;
        for (int d = 0; d < 3; ++d) c.bu[d] = m_bu[d];
    } else {
        c.a = m_a0 ? &m_lev[l].a : nullptr;
        for (int d = 0; d < 3; ++d) c.b[d] = &m_lev[l].b[d];
        // constant coefficients stay constant under the face average: the coarse levels run the constant-coefficient kernel forms too (their
        // arrays hold the same doubles: (x + x + x + x) 0.25 = x; the eta form keeps its one-component arrays, see prepare())
        if (m_buni_coarse) {
            c.b_uniform = 1;
            for (int d = 0; d < 3; ++d) c.bu[d] = m_bu[d];
            c.tensor_eta = m_tensor_eta ? 1 : 0;
        }
    }
    return c;
}

void CellMG::prepare()
{
    ProfScope ps_prof_("cmg_prepare");
    IAMRX_ASSERT(m_b0[0] && m_b0[1] && m_b0[2]);
    // singular <=> no 'a' term and no Dirichlet boundary (MLABecLaplacian::m_is_singular)
    m_singular = !(m_alpha != 0.0 && m_a0);
    for (int d = 0; d < 3; ++d)
        for (auto& b : m_bcn) if (!m_g.periodic[d] && (b.lo[d] == lo_dirichlet || b.hi[d] == lo_dirichlet)) m_singular = false;
    if (m_cf) m_singular = false;           // coarse/fine faces carry Dirichlet data
    // Strongly diagonally dominant operators (alpha a - beta div b grad with beta b / h^2 << alpha a: the Crank-Nicolson viscous and
    // diffusive solves at small nu dt / h^2): the Gauss-Seidel sweeps of the finest level alone contract the error by more than a
    // V-cycle needs to, so no coarse hierarchy is built and a "cycle" is m_dd_sweeps sweeps.  Bound of the Jacobi contraction:
    // q / (alpha min(a) + q), q = beta max(b) sum_d 2 / h_d^2; red-black Gauss-Seidel contracts by about its square per sweep.
    // Same converged answer (tests/test_gpu_sensitivity.py covers the solver choices); IAMRX_MG_DIAG_SHORTCUT=0 disables.
    // constant viscosity / diffusivity (every solve of a run without ns.variable_vel_visc / variable_scal_diff): the finest level's
    // kernels take the three constants instead of reading the face arrays; the coarser levels keep their averaged arrays
    m_buni = false;
    if (!m_sig && m_b0[0]->ncomp == 1) {
        m_buni = true;
        for (int d = 0; d < 3 && m_buni; ++d) {
            if (m_b0[d]->uniform_marked) {           // the owner's promise (MultiFab::mark_uniform): no scan
                m_bu[d] = m_b0[d]->uniform_value;
                if (tune("CHECK_UNIFORM", 0) != 0) {       // debugging aid: the promise is tested
                    double v = 0.0;
                    if (!mf_uniform_value(*m_b0[d], &v) || v != m_bu[d]) throw Error("iamrx: an array marked uniform is not (MultiFab::mark_uniform)");
                }
            }
            else m_buni = mf_uniform_value(*m_b0[d], &m_bu[d]);
        }
    }
    m_buni_coarse = m_buni && tune("MG_COARSE_UNIFORM", 1) != 0;
    m_dd_sweeps = 0;
    const bool dd_on = tune("MG_DIAG_SHORTCUT", 1) != 0;
    if (dd_on && m_alpha > 0.0 && m_beta > 0.0 && m_a0 && m_o.fixed_iters <= 0 && m_o.max_coarsening_level > 0) {
        double bmax = 0.0;
        for (int d = 0; d < 3; ++d) bmax = std::max(bmax, m_buni ? std::fabs(m_bu[d]) : m_b0[d]->norm0(0, m_b0[d]->ncomp, 0));
        if (m_tensor_eta) bmax *= 4.0 / 3.0;
        MultiFab inv(m_lev[0].layout, cell_type(), 1, 0);
        {
            const FabD *it = inv.d_tab, *at = m_a0->d_tab;
            for_each(*m_lev[0].layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
                const double a = at[f](i, j, k);
                it[f](i, j, k) = a > 0.0 ? 1.0 / a : 1.e300;
            });
        }
        const double ainv = inv.norm0(0, 1, 0);
        double q = 0.0;
        for (int d = 0; d < 3; ++d) q += 2.0 / (m_g.dx[d] * m_g.dx[d]);
        q *= m_beta * bmax;
        const double jac = ainv < 1.e299 ? q / (m_alpha / ainv + q) : 1.0;
        if (jac < 0.2) m_dd_sweeps = jac < 0.05 ? 2 : (jac < 0.12 ? 3 : 4);
        // red-black Gauss-Seidel contracts by about the square of the Jacobi factor per sweep: solve() sizes every cycle with it
        m_dd_rho = m_dd_sweeps > 0 ? std::max(jac * jac, 1.e-6) : 0.0;
        const int dd_force = (int)tune("MG_DD_SWEEPS", 0);
        if (m_dd_sweeps > 0 && dd_force > 0) { m_dd_sweeps = dd_force; m_dd_rho = 0.0; }
        if (m_o.verbose) printf("iamrx MLMG: Jacobi bound %.3e -> %d sweeps per cycle\n", jac, m_dd_sweeps);
    }
    // coarsen while every box is coarsenable (MLLinOp::defineGrids, mg_box_min_width = 2)
    m_lev.resize(1);
    while (m_dd_sweeps == 0 && (int)m_lev.size() <= m_o.max_coarsening_level) {
        Level& f = m_lev.back();
        // a single box of at most 8^3 cells is solved by the single-workgroup device bottom solver (k_abec_bottom): no need to coarsen
        // further (IAMRX_MG_DEVICE_BOTTOM=0: host-driven BiCGStab on the coarsest possible level, upstream's shape)
        if (m_o.device_bottom && !m_tensor && !m_o.bottom_smoother_only && abec_bottom_device_ok(f.g, *f.layout, m_bcn.data(), (int)m_bcn.size(), m_ncomp, m_cf)) break;
        bool dom_ok = true;
        for (int d = 0; d < 3; ++d) if (f.g.domain.len(d) % 2 != 0 || f.g.domain.len(d) / 2 < m_o.min_width) dom_ok = false;
        const bool iso = dom_ok && f.layout->coarsenable(2, m_o.min_width);
        const bool slab = !iso && mg_slab_level(f.g, *f.layout, m_o.min_width, m_o.slab != 0);
        if (!iso && !slab) break;
        Level c;
        c.g = f.g;
        if (slab) {
            c.g = mg_slab_geom(f.g);
            c.slab = true;
            c.virt = f.layout->coarsened(2);             // the one-plane coarsening the transfers write to
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
    const int nl = (int)m_lev.size();
    // several boxes covering the domain (a chopped level, the boxes of a sharded level): the one-launch sweep with its two-layer exchange
    {
        const bool has_a = m_a0 && m_alpha != 0.0;
        m_nbr = !m_cf && (m_sig || m_buni) && m_lev[0].layout->boxes.size() > 1 &&
                abec_gsrb_rb_nbr_level_ok(m_lev[0].g, *m_lev[0].layout, m_ncomp, m_sig != nullptr, has_a, (int)m_bcn.size(), m_bcn.data());
        m_sig2.clear(); m_a1.clear();
        if (m_nbr && m_sig) {
            m_sig2.define(m_lev[0].layout, cell_type(), 1, 2);
            // the caller's promise (as for the colour passes): one FACE ghost layer of the density filled, beyond domain walls too.  The
            // sweep also reads edge ghost cells -- beyond a wall AND behind a box-box face: what the box next door holds in its face ghost
            // cells there (it forms the same red update from them) -- so those are taken from the neighbour's copy, not from whatever the
            // caller's edge cells hold (poisoned edge cells must not change the answer)
            MultiFab::Copy(m_sig2, *m_sig, m_sig_comp, 0, 1, 1);
            m_sig2.FillBoundaryWallExt(m_lev[0].g, 1);
        }
        if (m_nbr && has_a) {
            m_a1.define(m_lev[0].layout, cell_type(), 1, 1);
            MultiFab::Copy(m_a1, *m_a0, 0, 0, 1, 0);
            m_a1.FillBoundary(m_lev[0].g);
        }
    }
    m_bottom_dev = m_o.device_bottom && m_dd_sweeps == 0 && !m_tensor && !m_o.bottom_smoother_only &&
                   abec_bottom_device_ok(m_lev.back().g, *m_lev.back().layout, m_bcn.data(), (int)m_bcn.size(), m_ncomp, m_cf);
    for (int l = 0; l < nl; ++l) {
        Level& L = m_lev[l];
        L.cor.define(L.layout, cell_type(), m_ncomp, (l == 0 && m_nbr) ? 2 : 1);
        L.res.define(L.layout, cell_type(), m_ncomp, (l == 0 && m_nbr) ? 1 : 0);
        L.rescor.define(L.layout, cell_type(), m_ncomp, 0);
        if (m_cf) {
            // the Dirichlet data sit half a coarse cell (of the AMR level below) behind the face on every MG level
            double loc[3];
            for (int d = 0; d < 3; ++d) loc[d] = 0.5 * m_ratio * m_g.dx[d];
            L.cftab = cf_make_tab(loc, L.g.dx, m_o.maxorder);
            L.cfm.define(L.layout, cell_type(), 1, l == 0 ? 2 : 1);
            cf_build_mask(L.g, L.cfm);
        }
        if (l > 0) {
            AbecCoef fc = coef(l - 1);
            if (L.agg) L.tmp_d.define(L.dist, cell_type(), m_ncomp, 0);
            if (L.slab) L.vres.define(L.virt, cell_type(), m_ncomp, 0);
            // (slab level: the transfer writes the one-plane virtual level, whose plane is duplicated into the level's two)
            auto coarsen_into = [&](MultiFab& dst, IndexType t, int nc, const std::function<void(MultiFab&)>& op) {
                LayoutP own = L.agg ? L.dist : L.layout;
                MultiFab dist_arr;
                MultiFab* target = &dst;
                if (L.agg) { dist_arr.define(own, t, nc, 0); target = &dist_arr; }
                if (L.slab) { MultiFab v(L.virt, t, nc, 0); op(v); slab_duplicate(*target, v); }
                else op(*target);
                if (L.agg) gather_to_replicated(dst, dist_arr);
            };
            if (m_a0) {
                L.a.define(L.layout, cell_type(), 1, 0);
                coarsen_into(L.a, cell_type(), 1, [&](MultiFab& c) { cc_restrict(c, *fc.a); });
            }
            for (int d = 0; d < 3; ++d) {
                // coarsening the finest level of an eta-form tensor operator: average the three-component coefficients it stands for
                MultiFab b3;
                const MultiFab* fb = fc.b[d];
                if (l == 1 && m_tensor_eta && !m_buni_coarse) { b3.define(m_lev[0].layout, face_type(d), 3, 0); tensor_bcoef(b3, *fc.b[d], d); fb = &b3; }
                L.b[d].define(L.layout, face_type(d), fb->ncomp, 0);
                coarsen_into(L.b[d], face_type(d), fb->ncomp, [&](MultiFab& c) { face_avgdown(c, *fb, d); });
            }
        }
    }
    bottom_direct_prepare();
}


// corners: also the edge / corner ghost cells that only the tensor cross terms read (not needed in front of a smoothing pass: the
// smoother acts on the 7-point part)
void CellMG::applyBC(int l, MultiFab& phi, bool inhomog, const MultiFab* bcval, bool corners)
{
    if (phi.ngrow > 1) { const int one[3] = {1, 1, 1}; phi.FillBoundary(m_lev[l].g, 0, phi.ncomp, one); }    // (the second layer is the sweep kernel's)
    else phi.FillBoundary(m_lev[l].g);
    // order: domain faces, coarse/fine faces (+ the frozen edge / corner values of the tensor operator), then the edge / corner cells
    // outside the physical domain, which are extrapolated from cells filled by the first two
    if (m_bcn.size() == 1) abec_apply_domain_bc(m_lev[l].g, phi, m_bcn[0], inhomog, bcval);
    else if (m_ncomp <= 3 && (int)m_bcn.size() >= m_ncomp)   // one BC per component (MLTensorOp::setDomainBC with per-component arrays, reference
        abec_apply_domain_bc_percomp(m_lev[l].g, phi, m_bcn.data(), m_ncomp, inhomog, bcval);     // Source/Diffusion.cpp:724-731): one launch
    else
        for (int n = 0; n < m_ncomp; ++n) abec_apply_domain_bc(m_lev[l].g, phi, m_bcn[n], inhomog, bcval, n, 1);
    if (m_cf) cf_fill_ghosts(phi, m_lev[l].cfm, m_lev[l].cftab, inhomog, bcval, m_tensor);
    if (m_tensor && corners) {
        if (m_bcn.size() == 1) fill_tensor_corners(m_lev[l].g, phi, m_bcn[0], inhomog, bcval);
        else for (int n = 0; n < m_ncomp; ++n) fill_tensor_corners(m_lev[l].g, phi, m_bcn[n], inhomog, bcval, n, 1);
    }
}

// level BC data at the coarse/fine ghost cells: the coarse solution interpolated along the faces
void CellMG::cf_bcval(MultiFab& bcval)
{
    if (!m_cf) return;
    LayoutP cl = m_lev[0].layout->coarsened(m_ratio);
    MultiFab cpatch(cl, cell_type(), m_ncomp, 1);
    cpatch.setVal(0.0);
    if (m_crse) parallel_copy(cpatch, *m_crse, 0, 0, m_ncomp, 0, 1, &m_cgeom, false);
    cf_interp_bndry(bcval, cpatch, m_lev[0].cfm, m_ratio);
    // tensor cross terms: the edge / corner coarse-fine ghost cells take the coarse data interpolated to their centres (no upstream
    // MLTensorOp::applyBCTensor to follow: DESIGN.md section 2)
    if (m_tensor) cf_interp_edges(bcval, cpatch, m_lev[0].cfm, m_ratio, m_cgeom);
}

static double dd_omega()
{
    return tune("MG_DD_OMEGA", 1.0);
}

// one-component coarse/fine levels: the colour passes keep the coarse/fine ghost cells current themselves (cf_maintain, k_abec.hip), so only
// the first pass of a smoothing call needs the k_cf_fill launch; IAMRX_CF_MAINTAIN=0: a fill in front of every pass
static bool cf_maintain_on()
{
    return tune("CF_MAINTAIN", 1) != 0;
}

bool CellMG::zero_first_pass_ok(int l, const MultiFab& sol) const
{
    if (m_dd_sweeps > 0 || fused_smoother_ok(l)) return false;
    AbecCoef c = coef(l);
    const bool wrap = !m_cf && periodic_wrap_ok(m_lev[l].g, *m_lev[l].layout, 2);
    // (walls applied inside the colour passes: the first pass from zero reads no ghost cell either)
    c.tensor = 0;
    const bool wk = !wrap && abec_gsrb_walls_inkernel_ok(m_lev[l].g, c, sol, (int)m_bcn.size(), m_bcn.data(), m_cf);
    return abec_gsrb_zero_ok(c, sol, (int)m_bcn.size(), wrap || wk, m_cf);
}

void CellMG::smooth(int l, MultiFab& sol, const MultiFab& rhs, bool skip_fill, bool cf_ghosts_current, bool sol_is_zero)
{
    AbecCoef c = coef(l);
    c.tensor = 0;   // the smoother acts on the ABec part; cross terms enter through the residual
    // one box spanning a fully periodic domain: the kernel reads the periodic images from the valid cells, no ghost fills
    const bool wrap = !m_cf && periodic_wrap_ok(m_lev[l].g, *m_lev[l].layout, 2);
    const bool maint = m_cf && m_ncomp == 1 && !m_tensor && cf_maintain_on();
    // one box spanning a domain with walls: the colour passes apply the wall conditions themselves (WallK, k_abec.hip) -- no k_abec_bc launch
    // in front of a pass; periodic directions (if any) keep their ghost fill
    if (m_lev[l].wk_flag < 0) m_lev[l].wk_flag = (!wrap && abec_gsrb_walls_inkernel_ok(m_lev[l].g, c, sol, (int)m_bcn.size(), m_bcn.data(), m_cf)) ? 1 : 0;
    const bool wk = m_lev[l].wk_flag == 1;
    const bool wk_per = wk && (m_lev[l].g.periodic[0] || m_lev[l].g.periodic[1] || m_lev[l].g.periodic[2]);
    for (int rb = 0; rb < 2; ++rb) {
        if (wk) { if (!skip_fill && wk_per) sol.FillBoundary(m_lev[l].g); }
        else if (!skip_fill && !wrap) {
            if (maint && (cf_ghosts_current || rb == 1)) {        // everything but the coarse/fine ghost cells
                sol.FillBoundary(m_lev[l].g);
                abec_apply_domain_bc(m_lev[l].g, sol, m_bcn[0], false, nullptr);
            } else applyBC(l, sol, false, nullptr, false);
        }
        // diagonally dominant shortcut: plain Gauss-Seidel -- over-relaxation leaves a (1 - omega) = 0.15 floor per sweep on an operator
        // that is almost its diagonal, where omega = 1 contracts by the square of the Jacobi factor
        abec_gsrb(m_lev[l].g, c, sol, rhs, rb, m_dd_sweeps > 0 ? dd_omega() : m_o.omega, m_bcn.data(), (int)m_bcn.size(), false, wrap, m_cf ? &m_lev[l].cfm : nullptr,
                  m_cf ? &m_lev[l].cftab : nullptr, maint, sol_is_zero && rb == 0, wk);
        skip_fill = false;
    }
}

// The fused sweep refreshes the ghost cells once per sweep, from a state in which the black cells off the box surfaces are
// already updated.  That equals the reference sequence (ghost fill in front of each colour) only if a ghost value depends on
// nothing but the first cell inside the box: periodic / neighbour images, Neumann, and Dirichlet extrapolation of order <= 2.
bool CellMG::fused_smoother_ok(int l) const
{
    // opt-in: on MI355X the single-pass kernel (62 B/cell of HBM traffic instead of 130) is still slower than the two colour
    // passes (233 VGPRs -> 2 waves/SIMD with three barriers per plane: 0.41 ms vs 2 x 0.18 ms at 256^3)
    const bool on = tune("GSRB_FUSED", 0) != 0;
    if (!on || m_cf) return false;
    const Level& L = m_lev[l];
    for (int d = 0; d < 3; ++d) {
        if (L.layout->max_len[d] < 16) return false;
        if (L.g.periodic[d]) continue;
        for (const auto& b : m_bcn)
            for (int side = 0; side < 2; ++side) {
                const int t = side == 0 ? b.lo[d] : b.hi[d];
                if (t == lo_neumann) continue;
                if (t == lo_dirichlet && std::min(L.g.domain.len(d) + 1, b.maxorder) <= 2) continue;
                return false;
            }
    }
    return true;
}

// smooth_n runs the multi-box sweep kernel on this level (the finest level of a hierarchy prepared for it, arrays with its ghost widths)
bool CellMG::nbr_sweep_ok(int l, const MultiFab& sol, const MultiFab& rhs) const
{
    if (l != 0 || !m_nbr || m_cf) return false;
    AbecCoef c = coef(l);
    c.tensor = 0;
    return abec_gsrb_rb_nbr_ok(m_lev[l].g, c, sol, rhs, (int)m_bcn.size(), m_bcn.data());
}

// smooth_n runs the sweep kernel with in-kernel coarse/fine faces on this level (a refined box strictly inside its domain, finest level)
bool CellMG::cf_sweep_ok(int l, const MultiFab& sol) const
{
    if (l != 0 || !m_cf) return false;
    AbecCoef c = coef(l);
    c.tensor = 0;
    return abec_gsrb_rb_cf_ok(m_lev[l].g, c, sol);
}

void CellMG::smooth_n(int l, MultiFab& sol, const MultiFab& rhs, int nsweeps, bool skip_first_fill, bool sol_is_zero, MultiFab* acc)
{
    if (l != 0) acc = nullptr;
    if (nsweeps <= 0) { if (sol_is_zero) sol.setVal(0.0); return; }
    if (cf_sweep_ok(l, sol)) {
        // red + black in one out-of-place launch per sweep, the coarse/fine ghost values formed inside the kernel (no k_cf_fill, no ghost
        // maintenance; sol's ghost cells are left stale: every reader behind a smoothing call fills them)
        AbecCoef c = coef(l);
        c.tensor = 0;
        Level& L = m_lev[l];
        if (!L.buf.defined()) L.buf.define(L.layout, cell_type(), m_ncomp, 1);
        MultiFab* a = &sol;
        MultiFab* b = &L.buf;
        if (sol_is_zero && (nsweeps & 1)) std::swap(a, b);
        const double om = m_dd_sweeps > 0 ? dd_omega() : m_o.omega;
        for (int i = 0; i < nsweeps; ++i) {
            const bool last = acc && i == nsweeps - 1;
            abec_gsrb_rb(L.g, c, *a, last ? *acc : *b, rhs, om, sol_is_zero && i == 0, m_bcn.data(), (int)m_bcn.size(), &L.cftab, last);
            if (last) { m_acc_done = true; return; }
            std::swap(a, b);
        }
        if (a != &sol) MultiFab::Copy(sol, *a, 0, 0, m_ncomp, 0);
        return;
    }
    if (!m_cf) {
        // one box spanning a periodic domain: red + black in one out-of-place launch per sweep (k_abec_gsrb_rb), ping-pong with the level's buffer
        AbecCoef c = coef(l);
        c.tensor = 0;
        // (finest level only: on a coarser level of 128 cells in x a march of a few planes does not beat two colour passes)
        if (l == 0 && abec_gsrb_rb_ok(m_lev[l].g, c, sol, (int)m_bcn.size(), m_bcn.data())) {
            Level& L = m_lev[l];
            if (!L.buf.defined()) L.buf.define(L.layout, cell_type(), m_ncomp, 1);
            MultiFab* a = &sol;
            MultiFab* b = &L.buf;
            // from a zero start the first sweep reads no input: an odd number of sweeps starts "from" the buffer and ends in sol without a copy
            if (sol_is_zero && (nsweeps & 1)) std::swap(a, b);
            const double om = m_dd_sweeps > 0 ? dd_omega() : m_o.omega;
            for (int i = 0; i < nsweeps; ++i) {
                const bool last = acc && i == nsweeps - 1;
                abec_gsrb_rb(L.g, c, *a, last ? *acc : *b, rhs, om, sol_is_zero && i == 0, m_bcn.data(), (int)m_bcn.size(), nullptr, last);
                if (last) { m_acc_done = true; return; }
                std::swap(a, b);
            }
            if (a != &sol) MultiFab::Copy(sol, *a, 0, 0, m_ncomp, 0);
            return;
        }
        // several boxes covering the domain: the same sweep per box, one two-layer exchange of the correction in front of it (none in
        // front of a sweep from zero), the ghost layer of the right-hand side once per V-cycle
        if (nbr_sweep_ok(l, sol, rhs)) {
            Level& L = m_lev[l];
            if (!L.buf.defined() || L.buf.ngrow != sol.ngrow) L.buf.define(L.layout, cell_type(), m_ncomp, sol.ngrow);
            if (!(&rhs == &L.res && L.res_filled)) {
                const int one[3] = {1, 1, 1};
                const_cast<MultiFab&>(rhs).FillBoundary(L.g, 0, m_ncomp, one);
                if (&rhs == &L.res) L.res_filled = true;
            }
            MultiFab* a = &sol;
            MultiFab* b = &L.buf;
            if (sol_is_zero && (nsweeps & 1)) std::swap(a, b);
            const double om = m_dd_sweeps > 0 ? dd_omega() : m_o.omega;
            // Overlap (IAMRX_HALO_OVERLAP, 1: on where the level exchanges with other ranks; 2: always; 0: off): the exchange of the two ghost
            // layers, k_abec_rb_ghost and the tiles next to box faces are issued on the context's side stream, the tiles that read no
            // ghost cell on the main stream in front of them -- the messages travel while the interior of the box is swept
            auto& ctx = Context::get();
            const int ov_mode = (int)tune("HALO_OVERLAP", 1);
            const CopyPlan& fplan = fill_boundary_plan(*L.layout, cell_type(), sol.ngrow, L.g);      // (built and uploaded in front of any fork)
            const bool overlap = ov_mode != 0 && (ov_mode == 2 || !fplan.peers.empty()) && abec_gsrb_rb_nbr_splits(L.g, *L.layout);
            for (int i = 0; i < nsweeps; ++i) {
                const bool z = sol_is_zero && i == 0;
                const bool last = acc && i == nsweeps - 1;
                MultiFab& out = last ? *acc : *b;
                if (z || !overlap) {
                    if (!z) a->FillBoundary(L.g);
                    abec_gsrb_rb_nbr(L.g, c, *a, out, rhs, om, z, m_bcn.data(), (int)m_bcn.size(), 0, nullptr, last);
                } else {
                    ctx.fork_side();
                    abec_gsrb_rb_nbr(L.g, c, *a, out, rhs, om, false, m_bcn.data(), (int)m_bcn.size(), 1, ctx.stream, last);
                    a->FillBoundary(L.g, 0, m_ncomp, nullptr, -1, ctx.side);
                    abec_gsrb_rb_nbr(L.g, c, *a, out, rhs, om, false, m_bcn.data(), (int)m_bcn.size(), 2, ctx.side, last);
                    ctx.join_side();
                }
                if (last) { m_acc_done = true; return; }
                std::swap(a, b);
            }
            if (a != &sol) MultiFab::Copy(sol, *a, 0, 0, m_ncomp, 0);
            return;
        }
    }
    if (!fused_smoother_ok(l)) {
        for (int i = 0; i < nsweeps; ++i) smooth(l, sol, rhs, skip_first_fill && i == 0, i > 0, sol_is_zero && i == 0);
        return;
    }
    IAMRX_ASSERT(!sol_is_zero);
    Level& L = m_lev[l];
    AbecCoef c = coef(l);
    c.tensor = 0;   // the smoother acts on the ABec part; cross terms enter through the residual
    if (!L.buf.defined()) L.buf.define(L.layout, cell_type(), m_ncomp, 1);
    MultiFab* a = &sol;
    MultiFab* b = &L.buf;
    for (int i = 0; i < nsweeps; ++i) {
        if (!(skip_first_fill && i == 0)) applyBC(l, *a, false, nullptr);
        abec_gsrb_fused(L.g, c, *a, *b, rhs, m_o.omega, m_bcn.data(), (int)m_bcn.size());
        applyBC(l, *b, false, nullptr);
        abec_gsrb(L.g, c, *b, rhs, 1, m_o.omega, m_bcn.data(), (int)m_bcn.size(), true);
        std::swap(a, b);
    }
    if (a != &sol) MultiFab::Copy(sol, *a, 0, 0, m_ncomp, 0);
}

void CellMG::subtract_mean(int l, MultiFab& mf)
{
    const double ncell = (double)m_lev[l].g.domain.npts();
    for (int n = 0; n < m_ncomp; ++n) {
        double s = mf.sum_unique(m_lev[l].g, n);
        mf_add_scalar(mf, -s / ncell, n, 1, 0);
    }
}

int CellMG::bicgstab(int l, MultiFab& sol, const MultiFab& rhs, double eps_rel, double eps_abs, int& niters)
{
    Level& L = m_lev[l];
    const Geometry& g = L.g;
    const int nc = m_ncomp;
    AbecCoef c = coef(l);
    MultiFab ph(L.layout, cell_type(), nc, 1), sh(L.layout, cell_type(), nc, 1);
    MultiFab sorig(L.layout, cell_type(), nc, 0), p(L.layout, cell_type(), nc, 0), r(L.layout, cell_type(), nc, 0);
    MultiFab s(L.layout, cell_type(), nc, 0), rh(L.layout, cell_type(), nc, 0), v(L.layout, cell_type(), nc, 0), t(L.layout, cell_type(), nc, 0);
    ph.setVal(0.0); sh.setVal(0.0);
    applyBC(l, sol, false, nullptr);
    abec_residual(g, c, r, sol, &rhs);
    MultiFab::Copy(sorig, sol, 0, 0, nc, 0);
    MultiFab::Copy(rh, r, 0, 0, nc, 0);
    sol.setVal(0.0);
    double rnorm = r.norm0(0, nc, 0);
    const double rnorm0 = rnorm;
    int ret = 0, nit = 1;
    double rho_1 = 0, alpha = 0, omega = 0;
    if (rnorm0 == 0 || rnorm0 < eps_abs) { niters = 0; MultiFab::Copy(sol, sorig, 0, 0, nc, 0); return 0; }
    // Krylov bound: an N-unknown system needs at most N iterations in exact arithmetic; more only chases round-off
    // (observed: ~175 iterations per V-cycle on a 2^3 level whose rhs is at round-off level).  Cap at 2N.
    const long nunk = (long)g.domain.npts() * nc;
    const int maxiter = (int)std::min<long>(m_o.bottom_maxiter, std::max<long>(8, 2 * nunk));
    if (tune("KRYLOV_DEVICE", 1) != 0 && (Context::get().comm->nranks == 1 || L.layout->replicated)) {
        // krylov.h: the same loop with its scalars on the device (one status word per iteration comes back, one iteration late)
        ret = bicgstab_device(*L.layout, cell_type(), nc, g, sol, r, rh, ph, sh, v, t, rnorm0, eps_rel, eps_abs, maxiter,
                              [&](MultiFab& out, MultiFab& in) { applyBC(l, in, false, nullptr); abec_residual(g, c, out, in, nullptr); }, nit, rnorm);
    } else
    for (; nit <= maxiter; ++nit) {
        double rho;
        { const MultiFab* xs[1] = {&rh}; const MultiFab* ys[1] = {&r}; reduce_dots(1, xs, ys, 0, nc, g, &rho); }
        if (rho == 0) { ret = 1; break; }
        if (nit == 1) MultiFab::Copy(p, r, 0, 0, nc, 0);
        else {
            const double beta = (rho / rho_1) * (alpha / omega);
            mf_lincomb(p, 1.0, p, -omega, v, 0, nc, 0);
            mf_lincomb(p, 1.0, r, beta, p, 0, nc, 0);
        }
        MultiFab::Copy(ph, p, 0, 0, nc, 0);
        applyBC(l, ph, false, nullptr);
        abec_residual(g, c, v, ph, nullptr);
        double rhTv;
        { const MultiFab* xs[1] = {&rh}; const MultiFab* ys[1] = {&v}; reduce_dots(1, xs, ys, 0, nc, g, &rhTv); }
        if (rhTv != 0) alpha = rho / rhTv; else { ret = 2; break; }
        mf_lincomb(sol, 1.0, sol, alpha, ph, 0, nc, 0);
        mf_lincomb(s, 1.0, r, -alpha, v, 0, nc, 0);
        rnorm = s.norm0(0, nc, 0);
        if (rnorm < eps_rel * rnorm0 || rnorm < eps_abs) break;
        MultiFab::Copy(sh, s, 0, 0, nc, 0);
        applyBC(l, sh, false, nullptr);
        abec_residual(g, c, t, sh, nullptr);
        double tv[2];
        { const MultiFab* xs[2] = {&t, &t}; const MultiFab* ys[2] = {&t, &s}; reduce_dots(2, xs, ys, 0, nc, g, tv); }
        if (tv[0] != 0) omega = tv[1] / tv[0]; else { ret = 3; break; }
        mf_lincomb(sol, 1.0, sol, omega, sh, 0, nc, 0);
        mf_lincomb(r, 1.0, s, -omega, t, 0, nc, 0);
        rnorm = r.norm0(0, nc, 0);
        if (rnorm < eps_rel * rnorm0 || rnorm < eps_abs) break;
        if (omega == 0) { ret = 4; break; }
        rho_1 = rho;
    }
    if (ret == 0 && rnorm > eps_rel * rnorm0 && rnorm > eps_abs) ret = 8;
    if ((ret == 0 || ret == 8) && rnorm < rnorm0) mf_lincomb(sol, 1.0, sol, 1.0, sorig, 0, nc, 0);
    else { sol.setVal(0.0); mf_lincomb(sol, 1.0, sol, 1.0, sorig, 0, nc, 0); }
    niters = nit;
    return ret;
}

// Krylov iterations of the device bottom solver, summed over the V-cycles of the running solve (one counter: solves do not overlap)
static int* bottom_iters_dev()
{
    static int* d = nullptr;
    if (!d) { IAMRX_HIP_CHECK(hipMalloc(&d, sizeof(int))); IAMRX_HIP_CHECK(hipMemset(d, 0, sizeof(int))); }
    return d;
}


// ---------------------------------------------------------------------------- direct bottom solve of the tensor operator
// The coarsest level of a tensor solve (MLTensorOp: three coupled components, cross terms) is 2^3 ... 3^3 cells.  amrex::MLMG hands it to
// BiCGStab (bottom_reltol 1e-4); driven from the host that is ~60 launches and ~10 read-backs per V-cycle (0.8 ms of a 3.8 ms cycle at
// 256^3, and the read-backs drain the launch queue).  Here the level's operator matrix is used instead: with constant viscosity the
// operator is A = alpha diag(a) + beta B, where B (the level's geometry, boundary conditions and viscosity constants only) is built ONCE
// per distinct level of a run -- column m = the operator kernels applied to the m-th unit vector, the same applyBC + abec_residual the
// Krylov iteration would call -- and kept on the device; a bottom solve is then one single-workgroup launch that assembles A from B and
// the level's a-term and solves A cor = res by Gauss-Jordan elimination with partial pivoting.  The bottom solve becomes exact instead of
// 1e-4: the same converged solution of the solve (every tensor test compares it with the oracle's BiCGStab hierarchy).
// IAMRX_TENSOR_BOTTOM_DIRECT (1): 0 = the host-driven BiCGStab.
namespace {
constexpr int DENSE_MAXN = 81;
// (the matrix is shared between the cache and every solver prepared with it: an entry evicted from the cache lives until the last
// CellMG that holds it is gone -- ADVICE round 5)
struct DenseEntry { std::vector<unsigned char> key; std::shared_ptr<double> dB; int N; };
std::vector<DenseEntry>& dense_cache() { static std::vector<DenseEntry> c; return c; }

__global__ void k_dense_set_one(FabD x, int i, int j, int k, int n, double v) { x(i, j, k, n) = v; }

__global__ void __launch_bounds__(256) k_dense_column(FabD y, BoxD b, int nc, double* col)
{
    const int nx = b.len(0), ny = b.len(1), ncell = nx * ny * b.len(2);
    for (int r = (int)threadIdx.x; r < ncell * nc; r += 256) {
        const int n = r / ncell, q = r - n * ncell;
        const int i = q % nx, j = (q / nx) % ny, k = q / (nx * ny);
        col[r] = y(b.lo[0] + i, b.lo[1] + j, b.lo[2] + k, n);
    }
}

// A cor = res with A = alpha diag(a) + beta B; unknown r = comp * ncell + cell (x fastest); one workgroup, the augmented matrix in LDS
__global__ void __launch_bounds__(256) k_dense_bottom(const double* __restrict__ B, int N, BoxD b, int nc, FabD a, int has_a, double alpha, double beta,
                                                     FabD res, FabD cor)
{
    extern __shared__ double M[];               // N rows of N + 1 entries
    __shared__ double F[DENSE_MAXN];
    __shared__ int piv;
    const int tid = (int)threadIdx.x, W = N + 1;
    const int nx = b.len(0), ny = b.len(1), ncell = nx * ny * b.len(2);
    auto cell = [&](int r, int& i, int& j, int& k, int& n) {
        n = r / ncell; const int q = r - n * ncell;
        i = b.lo[0] + q % nx; j = b.lo[1] + (q / nx) % ny; k = b.lo[2] + q / (nx * ny);
    };
    for (int idx = tid; idx < N * N; idx += 256) {
        const int c = idx / N, r = idx - c * N;     // B is column-major: consecutive threads read consecutive entries
        double v = beta * B[idx];
        if (r == c && has_a) { int i, j, k, n; cell(r, i, j, k, n); v += alpha * a(i, j, k, 0); }
        M[r * W + c] = v;
    }
    for (int r = tid; r < N; r += 256) { int i, j, k, n; cell(r, i, j, k, n); M[r * W + N] = res(i, j, k, n); }
    __syncthreads();
    for (int p = 0; p < N; ++p) {
        if (tid < 64) {                          // pivot row: the largest |M[r][p]|, r >= p (first wavefront)
            double best = -1.0; int bi = p;
            for (int r = p + tid; r < N; r += 64) { const double v = fabs(M[r * W + p]); if (v > best) { best = v; bi = r; } }
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) piv = bi;
        }
        __syncthreads();
        const int pr = piv;
        if (pr != p) for (int c = p + tid; c < W; c += 256) { const double t = M[p * W + c]; M[p * W + c] = M[pr * W + c]; M[pr * W + c] = t; }
        __syncthreads();
        const double d = M[p * W + p];
        for (int r = tid; r < N; r += 256) F[r] = r == p ? 0.0 : M[r * W + p] / d;
        __syncthreads();
        const int nc_left = W - (p + 1);
        for (int idx = tid; idx < N * nc_left; idx += 256) {
            const int r = idx / nc_left, c = p + 1 + (idx - r * nc_left);
            M[r * W + c] -= F[r] * M[p * W + c];
        }
        __syncthreads();
    }
    for (int r = tid; r < N; r += 256) { int i, j, k, n; cell(r, i, j, k, n); cor(i, j, k, n) = M[r * W + N] / M[r * W + r]; }
}
}  // namespace

void CellMG::bottom_direct_prepare()
{
    m_bottom_direct = false;
    const int l = (int)m_lev.size() - 1;
    if (tune("TENSOR_BOTTOM_DIRECT", 1) == 0 || !m_tensor || m_dd_sweeps > 0 || l < 1 || m_o.bottom_smoother_only || !m_buni_coarse) return;
    if (!(m_a0 && m_alpha != 0.0)) return;                   // (the tensor solves of the path all carry the density term: never singular)
    const Level& L = m_lev[l];
    if (L.slab || L.layout->boxes.size() != 1 || L.layout->nlocal() != 1) return;
    const BoxD b = L.layout->boxes[0];
    const int N = (int)b.npts() * m_ncomp;
    // the box is the whole level: it spans the domain, or (a refined level: homogeneous coarse/fine values on its faces, k_cf_fill) it is the only one
    if (N > DENSE_MAXN || (!m_cf && b.npts() != L.g.domain.npts()) || (m_cf && tune("TENSOR_BOTTOM_DIRECT_CF", 1) == 0)) return;
    // what B depends on
    std::vector<unsigned char> key;
    auto put = [&](const void* p, size_t n) { const unsigned char* q = (const unsigned char*)p; key.insert(key.end(), q, q + n); };
    put(L.g.domain.lo, sizeof(L.g.domain.lo)); put(L.g.domain.hi, sizeof(L.g.domain.hi)); put(L.g.dx, sizeof(L.g.dx)); put(L.g.periodic, sizeof(L.g.periodic));
    put(&m_ncomp, sizeof(int));
    for (const DomainBC& d : m_bcn) { put(d.lo, sizeof(d.lo)); put(d.hi, sizeof(d.hi)); put(&d.maxorder, sizeof(int)); }
    put(m_bu, sizeof(m_bu));
    const int te = m_tensor_eta ? 1 : 0;
    put(&te, sizeof(int));
    const int cf = m_cf ? 1 : 0;
    put(&cf, sizeof(int));
    if (m_cf) {
        // a box that touches no side of the domain: B depends on its size only, not on where it lies (regrids move it)
        bool inside = true;
        for (int d = 0; d < 3; ++d) inside = inside && b.lo[d] > L.g.domain.lo[d] && b.hi[d] < L.g.domain.hi[d];
        const int len[3] = {b.len(0), b.len(1), b.len(2)};
        if (inside) put(len, sizeof(len)); else { put(b.lo, sizeof(b.lo)); put(b.hi, sizeof(b.hi)); }
        put(L.cftab.c, sizeof(L.cftab.c)); put(&L.cftab.maxorder, sizeof(int));
    }
    for (const DenseEntry& e : dense_cache()) if (e.key == key) { m_dBh = e.dB; m_dB = e.dB.get(); m_dN = e.N; m_bottom_direct = true; return; }
    // build: column m = the operator (alpha 0, beta 1) applied to the m-th unit vector under the level's homogeneous boundary conditions
    auto& ctx = Context::get();
    double* dB = nullptr;
    IAMRX_HIP_CHECK(hipMalloc(&dB, sizeof(double) * (size_t)N * N));
    MultiFab x(L.layout, cell_type(), m_ncomp, 1), y(L.layout, cell_type(), m_ncomp, 0);
    AbecCoef c = coef(l);
    c.alpha = 0.0; c.beta = 1.0; c.a = nullptr;
    const int nx = b.len(0), ny = b.len(1), ncell = (int)b.npts();
    for (int m = 0; m < N; ++m) {
        const int n = m / ncell, q = m - n * ncell;
        x.setVal(0.0);
        hipLaunchKernelGGL(k_dense_set_one, dim3(1), dim3(1), 0, ctx.stream, x.h_tab[0], b.lo[0] + q % nx, b.lo[1] + (q / nx) % ny, b.lo[2] + q / (nx * ny), n, 1.0);
        applyBC(l, x, false, nullptr);
        abec_residual(L.g, c, y, x, nullptr);
        hipLaunchKernelGGL(k_dense_column, dim3(1), dim3(256), 0, ctx.stream, y.h_tab[0], b, m_ncomp, dB + (size_t)m * N);
    }
    ctx.sync();
    if (dense_cache().size() >= 64) dense_cache().erase(dense_cache().begin());      // (many distinct coarsest levels: the oldest entry leaves the cache)
    std::shared_ptr<double> h(dB, [](double* q) { (void)hipFree(q); });
    dense_cache().push_back(DenseEntry{key, h, N});
    m_dBh = h; m_dB = dB; m_dN = N; m_bottom_direct = true;
}

void CellMG::bottom_direct_solve()
{
    const int l = (int)m_lev.size() - 1;
    Level& L = m_lev[l];
    const AbecCoef c = coef(l);
    const BoxD b = L.layout->boxes[0];
    const size_t lds = sizeof(double) * (size_t)m_dN * (m_dN + 1);
    hipLaunchKernelGGL(k_dense_bottom, dim3(1), dim3(256), lds, Context::get().stream, m_dB, m_dN, b, m_ncomp, c.a->h_tab[0], 1, c.alpha, c.beta,
                       L.res.h_tab[0], L.cor.h_tab[0]);
}

void CellMG::bottom_solve(MGStats& st)
{
    ProfScope ps_prof_("cmg_bottom");
    const int l = (int)m_lev.size() - 1;
    Level& L = m_lev[l];
    if (m_dd_sweeps > 0) {                 // diagonally dominant operator: no hierarchy, see prepare()
        AbecCoef c = coef(l);
        c.tensor = 0;
        if ((!m_cf && (abec_gsrb_rb_ok(L.g, c, L.cor, (int)m_bcn.size(), m_bcn.data()) ||
                       nbr_sweep_ok(l, L.cor, L.res))) || cf_sweep_ok(l, L.cor)) {
            smooth_n(l, L.cor, L.res, m_dd_sweeps, true, true, m_acc);      // the first sweep takes the correction as zero: no fill, nothing read
            return;
        }
        L.cor.setVal(0.0);
        smooth_n(l, L.cor, L.res, m_dd_sweeps, true);
        return;
    }
    if (m_bottom_direct) { bottom_direct_solve(); return; }
    L.cor.setVal(0.0);
    if (m_o.bottom_smoother_only) {
        smooth_n(l, L.cor, L.res, m_o.nuf, true);
        return;
    }
    if (m_bottom_dev) {
        const long nunk = L.layout->total_cells() * m_ncomp;
        const int maxiter = (int)std::min<long>(m_o.bottom_maxiter, std::max<long>(8, 2 * nunk));
        abec_bottom_solve(L.g, coef(l), L.cor, L.res, m_bcn[0], m_singular, m_o.bottom_reltol, maxiter, m_o.nub, m_o.nuf, m_o.omega, bottom_iters_dev(),
                          m_cf ? &L.cftab : nullptr);
        return;
    }
    MultiFab b(L.layout, cell_type(), m_ncomp, 0);
    MultiFab::Copy(b, L.res, 0, 0, m_ncomp, 0);
    if (m_singular) subtract_mean(l, b);
    int nit = 0;
    int ret = bicgstab(l, L.cor, b, m_o.bottom_reltol, -1.0, nit);
    st.bottom_iters_total += nit;
    if (ret != 0) {
        L.cor.setVal(0.0);
        smooth_n(l, L.cor, L.res, m_o.nuf, true);
    }
    const int nn = (ret == 0) ? m_o.nub : m_o.nuf;
    smooth_n(l, L.cor, L.res, nn, false);
}

// the last two levels run as one launch (k_abec_tail): the bottom level is the device bottom solver's and the level above it a single box of
// at most 16^3 cells with the same kind of faces; no agglomeration / slab transfer between the two
bool CellMG::tail_fused() const
{
    const int nl = (int)m_lev.size();
    if (nl < 2 || !m_bottom_dev || m_cf || m_dd_sweeps > 0 || m_o.nu1 <= 0) return false;
    const Level& F = m_lev[nl - 2];
    const Level& C = m_lev[nl - 1];
    if (C.agg || C.slab || F.slab || fused_smoother_ok(nl - 2)) return false;
    AbecCoef cF = coef(nl - 2);
    return abec_tail_ok(F.g, *F.layout, C.g, *C.layout, cF, m_bcn.data(), (int)m_bcn.size(), m_ncomp);
}

void CellMG::vcycle(MGStats& st)
{
    const int nl = (int)m_lev.size();
    m_lev[0].res_filled = false;
    const bool tail = tail_fused();
    const int nsm = tail ? nl - 2 : nl - 1;          // levels smoothed by the loops below
    for (int l = 0; l < nsm; ++l) {
        Level& L = m_lev[l];
        // zero initial guess of the correction: where the first colour pass reads no ghost cell it also takes the place of the fill
        // (the sweep kernel takes "zero" per component: the three components of a tensor solve start from zero too)
        bool rb_zero = false;
        if (l == 0 && !m_cf && m_ncomp > 1) { AbecCoef cz = coef(l); cz.tensor = 0; rb_zero = abec_gsrb_rb_ok(L.g, cz, L.cor, (int)m_bcn.size(), m_bcn.data()); }
        const bool z = m_o.nu1 > 0 && (zero_first_pass_ok(l, L.cor) || nbr_sweep_ok(l, L.cor, L.res) || cf_sweep_ok(l, L.cor) || rb_zero);
        if (!z) L.cor.setVal(0.0);
        smooth_n(l, L.cor, L.res, m_o.nu1, true, z);
        const AbecCoef cl = coef(l);
        // (one box spanning a periodic domain: the fused residual + restriction reads the periodic images itself)
        if (m_cf || !abec_residual_reads_no_ghosts(L.g, cl, L.rescor, L.cor, L.res, true)) applyBC(l, L.cor, false, nullptr);
        // the restriction writes: the coarse level's residual; its distributed form (agglomerated level: gathered afterwards); or, for a
        // slab level, the one-plane virtual level, whose plane is then duplicated
        Level& C = m_lev[l + 1];
        MultiFab& held = C.agg ? C.tmp_d : C.res;
        MultiFab& target = C.slab ? C.vres : held;
        if (abec_resid_restrict_ok(cl, L.cor, L.res)) abec_resid_restrict(L.g, cl, target, L.cor, L.res);       // residual and restriction in one pass
        else {
            level_residual(L.g, cl, L.rescor, L.cor, &L.res);
            cc_restrict(target, L.rescor);
        }
        if (C.slab) slab_duplicate(held, C.vres);
        if (C.agg) gather_to_replicated(C.res, C.tmp_d);
    }
    if (tail) {
        Level& F = m_lev[nl - 2];
        Level& C = m_lev[nl - 1];
        const long nunk = C.layout->total_cells() * m_ncomp;
        const int maxiter = (int)std::min<long>(m_o.bottom_maxiter, std::max<long>(8, 2 * nunk));
        AbecCoef cF = coef(nl - 2), cC = coef(nl - 1);
        cF.tensor = 0; cC.tensor = 0;
        abec_tail_solve(F.g, cF, F.cor, F.res, C.g, cC, m_bcn[0], m_singular, m_o.bottom_reltol, maxiter, m_o.nub, m_o.nuf, m_o.nu1, m_o.nu2, m_o.omega,
                        bottom_iters_dev());
    } else
    bottom_solve(st);
    for (int l = nsm - 1; l >= 0; --l) {
        Level& L = m_lev[l];
        if (m_lev[l + 1].agg) {
            scatter_from_replicated(m_lev[l + 1].tmp_d, m_lev[l + 1].cor, 0);
            cc_prolong_add(L.cor, m_lev[l + 1].tmp_d);
        } else
        cc_prolong_add(L.cor, m_lev[l + 1].cor);
        smooth_n(l, L.cor, L.res, m_o.nu2, false, false, l == 0 ? m_acc : nullptr);
    }
}

void CellMG::apply(MultiFab& out, MultiFab& phi)
{
    const Geometry& g0 = m_lev[0].g;
    if (m_cf || !(g0.periodic[0] && g0.periodic[1] && g0.periodic[2])) {
        MultiFab bcval(m_lev[0].layout, cell_type(), m_ncomp, 1);
        MultiFab::Copy(bcval, phi, 0, 0, m_ncomp, 1);
        cf_bcval(bcval);
        applyBC(0, phi, true, &bcval);
    } else applyBC(0, phi, true, nullptr);         // fully periodic, no coarse/fine faces: no boundary data to keep
    level_residual(m_lev[0].g, coef(0), out, phi, nullptr);
}

void CellMG::fluxes(MultiFab& phi, MultiFab* const flux[3], MultiFab* const add_to[3])
{
    abec_flux(m_lev[0].g, coef(0), phi, flux, add_to);
}

MGStats CellMG::solve(MultiFab& phi, const MultiFab& rhs_in, double rtol, double atol)
{
    ProfScope ps_prof_("cmg_solve");
    auto& ctx = Context::get();
    MGStats st;
    st.nlevels = (int)m_lev.size();
    Level& L0 = m_lev[0];
    const int nc = m_ncomp;
    // the right-hand side is only changed (mean removed) for a singular system: everything else solves on the caller's array
    MultiFab rhs_own;
    if (m_singular) {
        rhs_own.define(L0.layout, cell_type(), nc, 0);
        MultiFab::Copy(rhs_own, rhs_in, 0, 0, nc, 0);
        subtract_mean(0, rhs_own);
    }
    const MultiFab& rhs = m_singular ? rhs_own : rhs_in;
    // boundary data = the ghost cells of the initial phi (domain faces) and the coarse solution (coarse/fine faces); a fully periodic
    // level without coarse/fine faces has none
    const bool has_bcdata = m_cf || !(L0.g.periodic[0] && L0.g.periodic[1] && L0.g.periodic[2]);
    MultiFab bcval_own;
    if (has_bcdata) {
        bcval_own.define(L0.layout, cell_type(), nc, 1);
        MultiFab::Copy(bcval_own, phi, 0, 0, nc, 1);
        cf_bcval(bcval_own);
    }
    const MultiFab* bcvp = has_bcdata ? &bcval_own : nullptr;

    applyBC(0, phi, true, bcvp);
    level_residual(L0.g, coef(0), L0.res, phi, &rhs, &st.resnorm0);      // the residual launch reduces its own max norm
    st.rhsnorm0 = rhs.norm0(0, nc, 0);
    const double max_norm = st.rhsnorm0 >= st.resnorm0 ? st.rhsnorm0 : st.resnorm0;
    const double res_target = std::max(atol, std::max(rtol, 1.e-16) * max_norm);
    st.resnorm = st.resnorm0;
    if (m_o.verbose) printf("iamrx MLMG: rhs %.6e resid0 %.6e target %.3e levels %d\n", st.rhsnorm0, st.resnorm0, res_target, st.nlevels);
    double vc_ms = 0.0;
    hipGraphExec_t vc_exec = nullptr;
    cycle_timer().used = 0;
    if (m_bottom_dev) IAMRX_HIP_CHECK(hipMemsetAsync(bottom_iters_dev(), 0, sizeof(int), ctx.stream));
    if (m_o.fixed_iters <= 0 && st.resnorm0 <= res_target) st.converged = 1;
    else {
        const int maxit = m_o.fixed_iters > 0 ? m_o.fixed_iters : m_o.max_iters;
        for (int iter = 0; iter < maxit; ++iter) {
            // singular system: the residual of a compatible right-hand side has zero mean up to round-off (the operator's columns sum to
            // zero), which amrex::MLMG removes in front of every cycle.  On a fully periodic level it is not removed here (the bottom solver removes its
            // own): a reduction, a host read-back and a pass over the level per iteration (0.13 ms of a 2.4 ms cycle at 256^3) for a
            // shift of the order of 1e-16 |rhs|.  IAMRX_MG_RES_MEAN=1 restores the per-iteration form.
            // (fully periodic level: the right-hand side has just lost its mean and no boundary data enters the residual -- nothing to remove)
            // A level with walls (boundary data present) keeps the per-iteration removal: there an incompatible component could build up
            // over many cycles and would not be projected out.
            if (m_singular && (has_bcdata || tune("MG_RES_MEAN", 0) != 0)) subtract_mean(0, L0.res);
            if (m_dd_sweeps > 0 && m_dd_rho > 0.0 && st.resnorm > 0.0) {
                // diagonally dominant operator: as many sweeps as the remaining reduction needs (measured at 256^3, nu dt/h^2 = 0.02:
                // 4 + 2 sweeps in two cycles instead of 3 x 3 sweeps; the second cycle only removes the lagged cross-term defect)
                const double need = std::min(0.5, res_target / st.resnorm);
                m_dd_sweeps = std::min(6, std::max(1, (int)std::ceil(std::log(need) / std::log(m_dd_rho))));
            }
            cycle_timer().mark(ctx.stream);
            // IAMRX_MG_GRAPH (0): the V-cycle of a hierarchy -- a fixed sequence of launches on fixed arrays, no host synchronisation, no
            // allocation once its lazy buffers exist -- is issued directly the first time (which defines those buffers), captured into a
            // hipGraph the second time and replayed from the third on: one graph launch instead of ~100 kernel launches per cycle
            const bool graph_ok = tune("MG_GRAPH", 0) != 0 && m_dd_sweeps == 0 && m_bottom_dev && ctx.comm->nranks == 1 && m_lev.size() > 1;
            if (graph_ok && iter >= 1) {
                if (!vc_exec) {
                    kernel_probes_pause(true);
                    hipGraph_t gr = nullptr;
                    IAMRX_HIP_CHECK(hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeThreadLocal));
                    vcycle(st);
                    IAMRX_HIP_CHECK(hipStreamEndCapture(ctx.stream, &gr));
                    IAMRX_HIP_CHECK(hipGraphInstantiate(&vc_exec, gr, nullptr, nullptr, 0));
                    IAMRX_HIP_CHECK(hipGraphDestroy(gr));
                    kernel_probes_pause(false);
                }
                IAMRX_HIP_CHECK(hipGraphLaunch(vc_exec, ctx.stream));
            } else {
            // IAMRX_MG_ACC_LAST_SWEEP (1): where the sweep kernel smooths the finest level its last sweep writes phi + correction into phi
            m_acc = (tune("MG_ACC_LAST_SWEEP", 1) != 0 && phi.ngrow >= 1 && phi.layout.get() == L0.layout.get() && phi.ncomp == nc) ? &phi : nullptr;
            m_acc_done = false;
            vcycle(st);
            m_acc = nullptr;
            }
            cycle_timer().mark(ctx.stream);
            if (!m_acc_done) mf_saxpy(phi, 1.0, L0.cor, 0, 0, nc, 0);
            m_acc_done = false;
            // (the ghost cells are filled once more behind the loop; a residual kernel that wraps its indices needs none here)
            if (m_cf || !abec_residual_reads_no_ghosts(L0.g, coef(0), L0.res, phi, rhs, false)) applyBC(0, phi, true, bcvp);
            level_residual(L0.g, coef(0), L0.res, phi, &rhs, &st.resnorm);
            st.iters = iter + 1;
            if (m_o.verbose) printf("iamrx MLMG: iter %d resid %.6e ratio %.3e\n", iter + 1, st.resnorm, st.resnorm / max_norm);
            if (m_o.fixed_iters <= 0 && st.resnorm <= res_target) { st.converged = 1; break; }
            if (!(st.resnorm < 1.e20 * max_norm)) throw Error("iamrx MLMG: failing to converge (residual blow-up)");
        }
        if (m_o.fixed_iters <= 0 && !st.converged) throw Error("iamrx MLMG: failed to converge after max_iters");
    }
    vc_ms = cycle_timer().total_ms();          // the last residual norm has synchronised the stream
    if (vc_exec) IAMRX_HIP_CHECK(hipGraphExecDestroy(vc_exec));
    if (st.iters > 0) st.vcycle_ms = vc_ms / st.iters;
    applyBC(0, phi, true, bcvp);
    if (m_bottom_dev && st.iters > 0) {
        int h = 0;
        IAMRX_HIP_CHECK(hipMemcpyAsync(&h, bottom_iters_dev(), sizeof(int), hipMemcpyDeviceToHost, ctx.stream));
        ctx.sync();
        st.bottom_iters_total = h;
    }
    return st;
}

}  // namespace iamrx
