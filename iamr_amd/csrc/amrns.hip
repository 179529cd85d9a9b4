// iamr_amd/csrc/amrns.hip -- the multi-level time step: subcycled advance of a hierarchy of NavierStokes levels, reflux,
// average down, MAC sync, sync registers, the multi-level (composite) nodal projections and the multi-level initialisation.
//
// Reference (all in /root/reference/Source):
//   Amr::timeStep / coarseTimeStep (upstream AMReX): advance(level); ncycle x timeStep(level+1); post_timestep(level)
//   NavierStokesBase::post_timestep        NavierStokesBase.cpp:2546-2636
//   NavierStokes::reflux                   NavierStokes.cpp:1736-1838
//   NavierStokes::avgDown / avgDown_StatePress NavierStokes.cpp:1845-1873, NavierStokesBase.cpp:4125-4163
//   NavierStokes::mac_sync                 NavierStokes.cpp:1438-1730
//   MacProj::mac_sync_solve / mac_sync_compute MacProj.cpp:359-479, 490-731
//   NavierStokesBase::level_sync           NavierStokesBase.cpp:1927-2044
//   Projection::MLsyncProject              Projection.cpp:457-607
//   SyncRegister                           SyncRegister.cpp:19-607
//   NavierStokesBase::SyncInterp           NavierStokesBase.cpp:3071-3276
//   Projection::initialVelocityProject / initialPressureProject / initialSyncProject Projection.cpp:615-1185
//   NavierStokes::post_init / post_init_press, NSB::post_init_state / post_init_estDT NavierStokes.cpp:1254-1432, NavierStokesBase.cpp:2307-2439
//   NavierStokesBase::computeNewDt         NavierStokesBase.cpp:945-1035
//
// MI355X-first choices: every register is a device-resident MultiFab driven by the cached copy plans of amr.hip (no per-face host
// loops); the composite nodal system is solved by a fast-adaptive-composite multigrid cycle whose level solves are the V-cycles of
// the hand-written nodal smoother / residual kernels (k_nodal.hip), with the coarse/fine coupling expressed through the same
// full-weighting restriction and trilinear interpolation kernels as the multigrid itself.
#include "operators.h"
#include "launch.h"
#include "amrns.h"
#include <chrono>
#include <cmath>
#include <algorithm>
#include <cstring>

namespace iamrx {

// ------------------------------------------------------------------------------------------------------------------------
// small level-wide helpers
namespace {

// cell MultiFab (1 ghost): 1 on the cells of `l` (neighbour boxes and periodic images included), 0 elsewhere
MultiFab coverage(const LayoutP& l, const Geometry& g)
{
    MultiFab cov(l, cell_type(), 1, 1);
    cov.setVal(0.0);
    mf_add_scalar(cov, 1.0, 0, 1, 0);
    cov.FillBoundary(g);
    return cov;
}
// cell MultiFab on the coarse layout (1 ghost): 1 on the coarse cells the fine layout covers
MultiFab fine_coverage(const LayoutP& crse, const LayoutP& fine, const Geometry& cgeom, int ratio)
{
    MultiFab fc(crse, cell_type(), 1, 1);
    fc.setVal(0.0);
    MultiFab ones(fine, cell_type(), 1, 0);
    ones.setVal(1.0);
    average_down(ones, fc, 0, 1, ratio);
    fc.FillBoundary(cgeom);
    return fc;
}
// node class with respect to a cell coverage (1 ghost): 0 none of the surrounding in-domain cells covered, 1 all of them, 2 some
void node_class(MultiFab& out, const MultiFab& cov, const Geometry& g)
{
    const FabD *ot = out.d_tab, *ct = cov.d_tab;
    const BoxD dom = g.domain;
    const int p0 = g.periodic[0], p1 = g.periodic[1], p2 = g.periodic[2];
    for_each(*out.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        int nin = 0, ntot = 0;
        for (int q = 0; q < 8; ++q) {
            const int ci = i - 1 + (q & 1), cj = j - 1 + ((q >> 1) & 1), ck = k - 1 + ((q >> 2) & 1);
            if ((!p0 && (ci < dom.lo[0] || ci > dom.hi[0])) || (!p1 && (cj < dom.lo[1] || cj > dom.hi[1])) || (!p2 && (ck < dom.lo[2] || ck > dom.hi[2]))) continue;
            ++ntot;
            nin += ct[f](ci, cj, ck) != 0.0;
        }
        ot[f](i, j, k) = nin == 0 ? 0.0 : (nin == ntot ? 1.0 : 2.0);
    });
}
}  // namespace
// y(ycomp..) *= (keep_where_zero ? (m == 0) : (m != 0)) on valid + ng
void mask_mult(MultiFab& y, int ycomp, int nc, const MultiFab& m, bool keep_where_zero, int ng)
{
    const FabD *yt = y.d_tab, *mt = m.d_tab;
    for_each(*y.layout, y.type, ng, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const bool on = keep_where_zero ? (mt[f](i, j, k) == 0.0) : (mt[f](i, j, k) != 0.0);
        if (!on) for (int n = 0; n < nc; ++n) yt[f](i, j, k, ycomp + n) = 0.0;
    });
}
namespace {
// node masks of the composite system: keep y where cls == v
void keep_class(MultiFab& y, const MultiFab& cls, double v)
{
    const FabD *yt = y.d_tab, *mt = cls.d_tab;
    for_each(*y.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        if (mt[f](i, j, k) != v) yt[f](i, j, k) = 0.0;
    });
}
DomainBC all_neumann(const Geometry& g)
{
    DomainBC b;
    for (int d = 0; d < 3; ++d) { b.lo[d] = g.periodic[d] ? lo_periodic : lo_neumann; b.hi[d] = b.lo[d]; }
    b.maxorder = 2;
    return b;
}
DomainBC op_bc(const DomainBC& bc)        // the operator does not distinguish inflow faces from walls
{
    DomainBC b = bc;
    for (int d = 0; d < 3; ++d) { if (b.lo[d] == lo_inflow) b.lo[d] = lo_neumann; if (b.hi[d] == lo_inflow) b.hi[d] = lo_neumann; }
    return b;
}
// fine nodal field -> nodes of the coarse level (full weighting = transpose of the trilinear interpolation / ratio^3, the weights
// of SyncRegister::FineAdd, SyncRegister.cpp:478-536).  fine: 1 ghost node layer, valid data set, ghosts arbitrary; out: coarse layout.
void restrict_to_crse(MultiFab& out, MultiFab& fine, const Geometry& fgeom, const Geometry& cgeom, int ratio)
{
    IAMRX_ASSERT(ratio == 2);
    fine.FillBoundary(fgeom);
    nodal_reflect_bc(fgeom, fine, all_neumann(fgeom));
    MultiFab cf(fine.layout->coarsened(ratio), node_type(), 1, 0);
    nodal_restrict(cf, fine);
    out.setVal(0.0);
    parallel_copy(out, cf, 0, 0, 1, 0, 0, &cgeom);
}

}  // namespace

void node_interp_from_crse(MultiFab& fine, const MultiFab& crse, const Geometry& cgeom, int ratio, const MultiFab* mask, bool add)
{
    if (fine.layout->boxes.empty()) return;
    MultiFab cp(fine.layout->coarsened(ratio), node_type(), 1, 0);
    cp.setVal(0.0);
    parallel_copy(cp, crse, 0, 0, 1, 0, 0, &cgeom);
    const FabD *ft = fine.d_tab, *ct = cp.d_tab;
    const FabD* mt = mask ? mask->d_tab : nullptr;
    const int r = ratio;
    const double rinv = 1.0 / (double)ratio;
    for_each(*fine.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        if (mt && mt[f](i, j, k) == 0.0) return;
        const int fi[3] = {i, j, k};
        int c0[3];
        double w[3];
        for (int d = 0; d < 3; ++d) {
            c0[d] = fi[d] >= 0 ? fi[d] / r : -((-fi[d] + r - 1) / r);
            w[d] = (double)(fi[d] - c0[d] * r) * rinv;
        }
        const FabD c = ct[f];
        double v = 0.0;
        for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
            const double ww = (cx ? w[0] : 1.0 - w[0]) * (cy ? w[1] : 1.0 - w[1]) * (cz ? w[2] : 1.0 - w[2]);
            if (ww != 0.0) v += ww * c(c0[0] + cx, c0[1] + cy, c0[2] + cz);
        }
        if (add) ft[f](i, j, k) += v; else ft[f](i, j, k) = v;
    });
}

void sync_interp_cellcons(MultiFab& dst, int dcomp, const MultiFab& crse, int scomp, int ncomp, const Geometry& cgeom, const Geometry& fgeom,
                          int ratio, const BCRec* bc)
{
    // FillPatchTwoLevels against an EMPTY fine level: every cell of dst is interpolated from the coarse data
    static LayoutP empty_layout;
    if (!empty_layout) empty_layout = std::make_shared<Layout>(std::vector<BoxD>{}, std::vector<int>{}, Context::get().comm->rank);
    MultiFab none(empty_layout, cell_type(), crse.ncomp, 0);
    MultiFab tmp(dst.layout, cell_type(), ncomp, 0);
    TimeData fd{nullptr, &none, 0.0, 0.0};
    TimeData cd{nullptr, &crse, 0.0, 0.0};
    double zero[24];
    std::memset(zero, 0, sizeof(zero));
    fillpatch_two_levels(tmp, 0, 0.0, fd, cd, scomp, ncomp, cgeom, fgeom, ratio, bc, zero, zero);      // HomExtDirFill
    MultiFab::Copy(dst, tmp, 0, dcomp, ncomp, 0);
}

void godunov_compute_aofs_sync(const Geometry& g, MultiFab& sync, int acomp, const MultiFab& S, int ncomp, const MultiFab* force,
                               const MultiFab* divu, MultiFab* const umac[3], MultiFab* const ucorr[3], const int* iconserv, double dt,
                               const BCRec* bc, bool is_velocity, bool use_forces_in_trans, MultiFab* const flux_out[3], int scheme)
{
    MultiFab scratch(S.layout, cell_type(), ncomp, 0);
    MultiFab ed[3];
    MultiFab* edp[3];
    for (int d = 0; d < 3; ++d) { ed[d].define(S.layout, face_type(d), ncomp, 0); edp[d] = &ed[d]; }
    godunov_compute_aofs(g, scratch, 0, S, ncomp, force, divu, umac, iconserv, dt, bc, is_velocity, use_forces_in_trans, edp, nullptr, scheme);
    auto& ctx = Context::get();
    for (int d = 0; d < 3; ++d) {                      // fluxes = edge state * Ucorr * area (NavierStokesBase.cpp:4681-4683)
        const double area = g.dx[(d + 1) % 3] * g.dx[(d + 2) % 3];
        const FabD *et = ed[d].d_tab, *ut = ucorr[d]->d_tab;
        const FabD* ft = flux_out ? flux_out[d]->d_tab : et;
        for_each(*S.layout, face_type(d), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
            const double u = ut[f](i, j, k);
            for (int n = 0; n < ncomp; ++n) ft[f](i, j, k, n) = et[f](i, j, k, n) * u * area;
        });
    }
    const FabD *fx = (flux_out ? flux_out[0] : &ed[0])->d_tab, *fy = (flux_out ? flux_out[1] : &ed[1])->d_tab, *fz = (flux_out ? flux_out[2] : &ed[2])->d_tab;
    const FabD* st = sync.d_tab;
    const double qvol = 1.0 / (g.dx[0] * g.dx[1] * g.dx[2]);
    for_each(*S.layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
        for (int n = 0; n < ncomp; ++n) {
            const double upd = -1.0 * qvol * ((fx[f](i + 1, j, k, n) - fx[f](i, j, k, n)) + (fy[f](i, j + 1, k, n) - fy[f](i, j, k, n))
                                            + (fz[f](i, j, k + 1, n) - fz[f](i, j, k, n)));
            st[f](i, j, k, acomp + n) -= upd;          // aofs -= update, update = -div F (:4826-4832)
        }
    });
}

// MacProj::mac_sync_compute, the form NavierStokes::mac_sync uses (Source/MacProj.cpp:488-731), on caller-owned arrays: the forcing of the
// velocity trace = gravity + viscous terms - grad p (divided by rho unless the momentum form is advected, :598-640), then ComputeAofs with
// is_sync = true for the velocities (-> Vsync) and for all scalars (-> Ssync).  Svel / Sscal: the FillPatch-ed state at prev_time with three
// ghost cells (Svel = rho u under do_mom_diff, :536-553; Sscal starts with the density); visc_vel (3 comps) / tf_scal (nscal comps: the
// scalars' forcing as :641-683 assemble it) with one ghost cell, null = zero.  The fluxes come back for the registers (:707-727, the caller's).
void mac_sync_compute(const Geometry& g, MultiFab* const ucorr[3], MultiFab& Vsync, MultiFab& Ssync, const MultiFab& Svel, const MultiFab& Sscal, int nscal,
                      const MultiFab* visc_vel, const MultiFab* tf_scal, const MultiFab& gradp, const MultiFab* divu, MultiFab* const umac[3],
                      const int* iconserv_scal, bool do_mom_diff, double gravity, double dt, const BCRec* bc_vel, const BCRec* bc_scal,
                      bool use_forces_in_trans, int scheme, MultiFab* const flux_vel[3], MultiFab* const flux_scal[3])
{
    auto& ctx = Context::get();
    const LayoutP& layout = Svel.layout;
    MultiFab tfv(layout, cell_type(), 3, 1), tfs0;
    if (!tf_scal) { tfs0.define(layout, cell_type(), nscal, 1); tfs0.setVal(0.0); tf_scal = &tfs0; }
    {
        const FabD *tt = tfv.d_tab, *gt = gradp.d_tab, *st = Sscal.d_tab, *vt = visc_vel ? visc_vel->d_tab : nullptr;
        const bool mom = do_mom_diff;
        const double grav = gravity;
        for_each(*layout, cell_type(), 1, ctx.stream, [=] __device__(int i, int j, int k, int fb) {
            const double rho = st[fb](i, j, k, 0);
            for (int n = 0; n < 3; ++n) {
                double t = ((fabs(grav) > 0.0001 && n == 2) ? grav * rho : 0.0) + (vt ? (double)vt[fb](i, j, k, n) : 0.0) - gt[fb](i, j, k, n);
                if (!mom) t /= rho;
                tt[fb](i, j, k, n) = t;
            }
        });
    }
    const int icv[3] = {do_mom_diff ? 1 : 0, do_mom_diff ? 1 : 0, do_mom_diff ? 1 : 0};
    godunov_compute_aofs_sync(g, Vsync, 0, Svel, 3, &tfv, divu, umac, ucorr, icv, dt, bc_vel, true, use_forces_in_trans, flux_vel, scheme);
    godunov_compute_aofs_sync(g, Ssync, 0, Sscal, nscal, tf_scal, divu, umac, ucorr, iconserv_scal, dt, bc_scal, false, use_forces_in_trans, flux_scal, scheme);
}

// MacProj::mac_sync_compute, the form with edge states handed in (Source/MacProj.cpp:733-786: ComputeAofs with known_edgestate and is_sync,
// one component): flux = edgestate(edge_comp) * Ucorr * area (NavierStokesBase.cpp:4681-4683), Sync(sync_indx) -= -div(flux) / vol
// (:4826-4832); flux_out (optional, one component) for the caller's advective flux register (FineAdd with the sync sign, :5083-5096)
void mac_sync_compute_edge(const Geometry& g, MultiFab* const ucorr[3], MultiFab& Sync, int sync_indx, MultiFab* const edgestate[3], int edge_comp,
                           MultiFab* const flux_out[3])
{
    auto& ctx = Context::get();
    const LayoutP& layout = Sync.layout;
    MultiFab fl[3];
    for (int d = 0; d < 3; ++d) {
        if (!flux_out) fl[d].define(layout, face_type(d), 1, 0);
        const double area = g.dx[(d + 1) % 3] * g.dx[(d + 2) % 3];
        const FabD *et = edgestate[d]->d_tab, *ut = ucorr[d]->d_tab, *ft = (flux_out ? flux_out[d] : &fl[d])->d_tab;
        const int ec = edge_comp;
        for_each(*layout, face_type(d), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) { ft[f](i, j, k) = et[f](i, j, k, ec) * ut[f](i, j, k) * area; });
    }
    const FabD *fx = (flux_out ? flux_out[0] : &fl[0])->d_tab, *fy = (flux_out ? flux_out[1] : &fl[1])->d_tab, *fz = (flux_out ? flux_out[2] : &fl[2])->d_tab;
    const FabD* st = Sync.d_tab;
    const double qvol = 1.0 / (g.dx[0] * g.dx[1] * g.dx[2]);
    const int sc = sync_indx;
    for_each(*layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
        const double upd = -1.0 * qvol * ((fx[f](i + 1, j, k) - fx[f](i, j, k)) + (fy[f](i, j + 1, k) - fy[f](i, j, k)) + (fz[f](i, j, k + 1) - fz[f](i, j, k)));
        st[f](i, j, k, sc) -= upd;
    });
}

// ------------------------------------------------------------------------------------------------------------------------
// sync residuals
MultiFab amr_sync_resid(NavierStokes& ns, const MultiFab& vold, const MultiFab& phi, const MultiFab& sig, bool crse_side, const MultiFab* rhcc)
{
    return sync_resid(ns.geom(), ns.lay(), ns.nodal_bc(), crse_side ? ns.fine->lay() : LayoutP(), crse_side ? ns.fine->ratio : 2, vold, phi, sig, rhcc);
}

MultiFab sync_resid(const Geometry& g, const LayoutP& layout, const DomainBC& bcn, const LayoutP& fine_layout, int fine_ratio, const MultiFab& vold,
                    const MultiFab& phi, const MultiFab& sig, const MultiFab* rhcc)
{
    auto& ctx = Context::get();
    const bool crse_side = (bool)fine_layout;
    MultiFab cls(layout, node_type(), 1, 0);
    MultiFab fc;
    MultiFab cnt = coverage(layout, g);                // the cells that count: cells of the level, not covered by the finer one
    if (crse_side) { fc = fine_coverage(layout, fine_layout, g, fine_ratio); node_class(cls, fc, g); mask_mult(cnt, 0, 1, fc, true, 1); }
    else node_class(cls, cnt, g);
    cc_mirror_bc(g, cnt);                               // mlndlap_fillbc_cc of the cell masks: mirrored across the non-periodic faces
    // sigma and velocity restricted to the cells that count (zero elsewhere, also in the ghost cells outside the level)
    MultiFab sm(layout, cell_type(), 1, 1), um(layout, cell_type(), 3, 1);
    sm.setVal(0.0); um.setVal(0.0);
    MultiFab::Copy(sm, sig, 0, 0, 1, 0);
    MultiFab::Copy(um, vold, 0, 0, 3, 0);
    if (crse_side) { mask_mult(sm, 0, 1, fc, true, 0); mask_mult(um, 0, 3, fc, true, 0); }
    sm.FillBoundary(g);
    cc_mirror_bc(g, sm);
    um.FillBoundary(g);
    {   // cells outside the physical domain keep the incoming values (inflow data; nodal_divu ignores the rest) where the cell they
        // mirror counts: the velocity is multiplied by the mirrored cell mask in compSyncResidualCoarse / Fine
        const FabD *ut = um.d_tab, *vt = vold.d_tab, *nt = cnt.d_tab;
        const BoxD dom = g.domain;
        const int p0 = g.periodic[0], p1 = g.periodic[1], p2 = g.periodic[2];
        for_each(*layout, cell_type(), 1, ctx.stream, [=] __device__(int i, int j, int k, int f) {
            const bool out = (!p0 && (i < dom.lo[0] || i > dom.hi[0])) || (!p1 && (j < dom.lo[1] || j > dom.hi[1])) || (!p2 && (k < dom.lo[2] || k > dom.hi[2]));
            if (out && nt[f](i, j, k) != 0.0) for (int n = 0; n < 3; ++n) ut[f](i, j, k, n) = vt[f](i, j, k, n);
        });
    }
    MultiFab rhs(layout, node_type(), 1, 0), ph(layout, node_type(), 1, 1), r(layout, node_type(), 1, 1);
    nodal_divu(g, rhs, um, 0, &bcn);
    if (rhcc) { MultiFab rc = make_rhcc(g, *rhcc, 0, 1.0, crse_side ? &fc : nullptr); nodal_rhcc_add(g, rhs, rc, bcn); }
    ph.setVal(0.0);
    MultiFab::Copy(ph, phi, 0, 0, 1, 0);
    ph.FillBoundary(g);
    nodal_reflect_bc(g, ph, op_bc(bcn));
    r.setVal(0.0);
    nodal_residual(g, r, ph, sm, &rhs);
    keep_class(r, cls, 2.0);
    // nodes on Dirichlet (outflow) faces carry no residual
    for (int d = 0; d < 3; ++d) {
        if (g.periodic[d]) continue;
        for (int side = 0; side < 2; ++side) {
            if ((side == 0 ? bcn.lo[d] : bcn.hi[d]) != lo_dirichlet) continue;
            const int face = side == 0 ? g.domain.lo[d] : g.domain.hi[d] + 1;
            const FabD* rt = r.d_tab;
            const int dd = d;
            for_each(*layout, node_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
                if ((dd == 0 ? i : (dd == 1 ? j : k)) == face) rt[f](i, j, k) = 0.0;
            });
        }
    }
    return r;
}

// ------------------------------------------------------------------------------------------------------------------------
// SyncRegister
namespace {
struct RegSlab { int fab; BoxD region; };
__global__ void __launch_bounds__(256) k_set_slabs(const RegSlab* __restrict__ slabs, const FabD* __restrict__ tab, double v)
{
    const RegSlab s = slabs[blockIdx.y];
    const FabD a = tab[s.fab];
    const int nx = s.region.len(0), ny = s.region.len(1);
    const long npts = s.region.npts();
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        const int i = s.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        a(i, s.region.lo[1] + (int)(r % ny), s.region.lo[2] + (int)(r / ny)) = v;
    }
}
}  // namespace

SyncRegister::SyncRegister(LayoutP fine, LayoutP crse, const Geometry& cgeom, const Geometry& fgeom, int ratio, const int phys_lo[3], const int phys_hi[3])
    : m_fine(std::move(fine)), m_crse(std::move(crse)), m_cgeom(cgeom), m_fgeom(fgeom), m_ratio(ratio)
{
    for (int d = 0; d < 3; ++d) { m_plo[d] = phys_lo[d]; m_phi[d] = phys_hi[d]; }
    m_reg.define(m_crse, node_type(), 1, 0);
    m_reg.setVal(0.0);
    m_onreg.define(m_crse, node_type(), 1, 0);
    m_vsfine.define(m_crse, node_type(), 1, 0);
    MultiFab fc = fine_coverage(m_crse, m_fine, m_cgeom, m_ratio);
    node_class(m_vsfine, fc, m_cgeom);
    // nodes on the faces of the coarsened fine boxes (and their periodic images)
    std::vector<BoxD> nb;
    for (auto& b : m_fine->boxes) {
        const BoxD cb = coarsen(b, m_ratio);
        for (int sz = -1; sz <= 1; ++sz) for (int sy = -1; sy <= 1; ++sy) for (int sx = -1; sx <= 1; ++sx) {
            const int sh[3] = {sx, sy, sz};
            bool ok = true;
            for (int d = 0; d < 3; ++d) if (sh[d] != 0 && !m_cgeom.periodic[d]) ok = false;
            if (!ok) continue;
            BoxD q = cb;
            for (int d = 0; d < 3; ++d) { q.lo[d] += sh[d] * m_cgeom.domain.len(d); q.hi[d] += sh[d] * m_cgeom.domain.len(d) + 1; }   // nodal box
            nb.push_back(q);
        }
    }
    // m_onreg = 1 on those nodes.  Rasterised: the six face slabs of every nodal box, clipped to every local coarse fab they reach, set by one
    // launch over the list of slabs (round 6: a loop over all boxes per coarse node took 0.4 s per register on a level of 431 boxes)
    auto& ctx = Context::get();
    m_onreg.setVal(0.0);
    std::vector<RegSlab> slabs;
    long maxpts = 1;
    for (int li = 0; li < m_onreg.nlocal(); ++li) {
        const BoxD vb = m_onreg.validbox(li);
        for (const BoxD& q : nb) {
            const BoxD in = intersect(q, vb);
            if (!in.ok()) continue;
            for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
                const int face = side == 0 ? q.lo[d] : q.hi[d];
                if (face < in.lo[d] || face > in.hi[d]) continue;
                BoxD r = in; r.lo[d] = r.hi[d] = face;
                slabs.push_back(RegSlab{li, r});
                maxpts = std::max(maxpts, r.npts());
            }
        }
    }
    if (!slabs.empty()) {
        RegSlab* d_s = (RegSlab*)ctx.alloc(slabs.size() * sizeof(RegSlab));
        IAMRX_HIP_CHECK(hipMemcpyAsync(d_s, slabs.data(), slabs.size() * sizeof(RegSlab), hipMemcpyHostToDevice, ctx.stream));
        const unsigned nbk = (unsigned)std::min<long>(64, (maxpts + 255) / 256);
        // (gridDim.y is limited to 65535: chunks of slabs)
        for (size_t s0 = 0; s0 < slabs.size(); s0 += 65535) {
            const unsigned ns = (unsigned)std::min<size_t>(65535, slabs.size() - s0);
            hipLaunchKernelGGL(k_set_slabs, dim3(nbk, ns), dim3(256), 0, ctx.stream, d_s + s0, m_onreg.d_tab, 1.0);
        }
        ctx.sync();
        ctx.free(d_s);
    }
}

// SyncRegister::CompAdd (SyncRegister.cpp:302-348): the residual of a sync projection on the levels above, formed on the fine side of THIS
// register's interface, is zeroed on the nodes of the next finer level's boxes (Pgrids: that level's boxes in the index space of the
// residual, periodic images included) and then added like a fine residual
void SyncRegister::CompAdd(MultiFab& sync_resid_fine, const Geometry& fgeom, const LayoutP& finer, int finer_ratio, double mult)
{
    MultiFab vsf(sync_resid_fine.layout, node_type(), 1, 0);
    MultiFab fc = fine_coverage(sync_resid_fine.layout, finer, fgeom, finer_ratio);
    node_class(vsf, fc, fgeom);
    mask_mult(sync_resid_fine, 0, 1, vsf, true, 0);
    FineAdd(sync_resid_fine, mult);
}

void SyncRegister::CrseInit(const MultiFab& resid, double mult)
{
    const FabD *rt = m_reg.d_tab, *ot = m_onreg.d_tab, *st = resid.d_tab;
    for_each(*m_crse, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        rt[f](i, j, k) = ot[f](i, j, k) != 0.0 ? mult * st[f](i, j, k) : 0.0;
    });
}

void SyncRegister::FineAdd(const MultiFab& resid_fine, double mult)
{
    MultiFab rf(m_fine, node_type(), 1, 1);
    rf.setVal(0.0);
    MultiFab::Copy(rf, resid_fine, 0, 0, 1, 0);
    mf_mult(rf, mult, 0, 1, 0);
    MultiFab rc(m_crse, node_type(), 1, 0);
    restrict_to_crse(rc, rf, m_fgeom, m_cgeom, m_ratio);
    const FabD *rt = m_reg.d_tab, *ot = m_onreg.d_tab, *ct = rc.d_tab;
    for_each(*m_crse, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        if (ot[f](i, j, k) != 0.0) rt[f](i, j, k) += ct[f](i, j, k);
    });
}

void SyncRegister::InitRHS(MultiFab& rhs)
{
    const FabD *rt = m_reg.d_tab, *ot = m_onreg.d_tab, *vt = m_vsfine.d_tab, *ht = rhs.d_tab;
    const BoxD dom = m_cgeom.domain;
    int olo[3], ohi[3];
    for (int d = 0; d < 3; ++d) { olo[d] = (!m_cgeom.periodic[d] && m_plo[d] == phys_outflow) ? 1 : 0; ohi[d] = (!m_cgeom.periodic[d] && m_phi[d] == phys_outflow) ? 1 : 0; }
    const int a0 = olo[0], a1 = olo[1], a2 = olo[2], b0 = ohi[0], b1 = ohi[1], b2 = ohi[2];
    for_each(*m_crse, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        double v = ot[f](i, j, k) != 0.0 ? rt[f](i, j, k) : 0.0;
        if ((a0 && i == dom.lo[0]) || (b0 && i == dom.hi[0] + 1) || (a1 && j == dom.lo[1]) || (b1 && j == dom.hi[1] + 1) || (a2 && k == dom.lo[2]) || (b2 && k == dom.hi[2] + 1)) v = 0.0;
        if (vt[f](i, j, k) == 1.0) v = 0.0;             // bndry_mask: nodes surrounded by fine cells only
        ht[f](i, j, k) = v;
    });
}

NavierStokes::~NavierStokes() = default;

// ------------------------------------------------------------------------------------------------------------------------
// Composite nodal projection over levels c0 .. c0+nl-1 (Hydro::NodalProjector::project on several AMR levels as driven by
// Projection::doMLMGNodalProjection, Projection.cpp:2385-2567).
//
// Discretisation = the conforming Q1 finite-element composite operator: the cells of a level
// that the next finer level does not cover contribute their element matrices; a node of level l+1 on that level's boundary is a
// slave (trilinear interpolant of level l), what the fine cells contribute to it goes to level l with the transposed weights /
// ratio^3 (full-weighting restriction); boundary nodes of the coarsest level of the solve and nodes on Dirichlet faces keep their
// incoming value.  Solver: fast adaptive composite cycle -- a symmetric sweep finest -> coarsest -> finest of level corrections; each
// is one V-cycle of that level's own multigrid (zero data on the level's boundary) on the composite residual restricted to the
// level (full weighting of the finer levels' residuals), its result interpolated trilinearly to the unknowns of every finer level
// -- repeated until the composite residual meets MLMG's criterion.
namespace {

struct CLev {
    const ProjLevel* pl;
    Geometry g;
    LayoutP layout;
    std::unique_ptr<NodalMG> mg;
    MultiFab sigm;          // sigma on the uncovered cells, ghost cells filled
    MultiFab own, slave;    // node masks (1 / 0)
    MultiFab fcov;          // coverage by the next level of the solve (cells, 1 ghost)
    MultiFab b, x, r, y, e; // node arrays, 1 ghost
    MultiFab bb;            // scratch node array, 1 ghost node that stays zero: what a level hands down (the right-hand side of a level's correction
                            // is written into its solver's own residual array)
    DomainBC bc;            // operator BC
    double wscale;
};

void fill_nodes(CLev& L, MultiFab& x) { x.FillBoundary(L.g); nodal_reflect_bc(L.g, x, L.bc); }

void comp_fill_slaves(std::vector<CLev>& L)
{
    for (size_t l = 0; l < L.size(); ++l) {
        if (l > 0) node_interp_from_crse(L[l].x, L[l - 1].x, L[l - 1].g, L[l].pl->ratio, &L[l].slave, false);
        fill_nodes(L[l], L[l].x);
    }
}

// y = A x on the unknowns (0 elsewhere) of the levels >= lmin (the rows of a level need the contributions of the finer levels only);
// with_r: r = b - y as well.  One pass per level over the node arrays: y_total = (contribution of the finer level) + A x, the slave rows
// of y_total go down to the coarser level (bb), y = y_total on the unknowns, 0 elsewhere
void comp_apply(std::vector<CLev>& L, int lmin = 0, bool with_r = false)
{
    comp_fill_slaves(L);
    const int nl = (int)L.size();
    for (int l = nl - 1; l >= lmin; --l) {
        MultiFab ax(L[l].layout, node_type(), 1, 0);
        nodal_residual(L[l].g, ax, L[l].x, L[l].sigm, nullptr);
        const bool has_in = l < nl - 1, down = l > lmin;
        const FabD *at = ax.d_tab, *yt = L[l].y.d_tab, *ot = L[l].own.d_tab, *st = L[l].slave.d_tab, *bt = L[l].bb.d_tab, *rt = L[l].r.d_tab, *ft = L[l].b.d_tab;
        for_each(*L[l].layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
            const double yy = has_in ? yt[f](i, j, k) + at[f](i, j, k) : (double)at[f](i, j, k);
            if (down) bt[f](i, j, k) = st[f](i, j, k) != 0.0 ? yy : 0.0;
            const double yo = ot[f](i, j, k) != 0.0 ? yy : 0.0;
            yt[f](i, j, k) = yo;
            if (with_r) rt[f](i, j, k) = 1.0 * ft[f](i, j, k) + -1.0 * yo;
        });
        if (down) restrict_to_crse(L[l - 1].y, L[l].bb, L[l].g, L[l - 1].g, L[l].pl->ratio);
    }
}

double comp_dot_own(std::vector<CLev>& L, int which /*0: <own,b>, 1: <own,own>, 2: <own,x>*/)
{
    double s = 0.0;
    for (auto& l : L) {
        const MultiFab* xs[1] = {&l.own};
        const MultiFab* ys[1] = {which == 0 ? &l.b : (which == 2 ? &l.x : &l.own)};
        double v;
        Geometry gg = l.g;
        for (int d = 0; d < 3; ++d) { gg.half_lo[d] = (!gg.periodic[d] && l.bc.lo[d] == lo_neumann) ? 1 : 0; gg.half_hi[d] = (!gg.periodic[d] && l.bc.hi[d] == lo_neumann) ? 1 : 0; }
        reduce_dots(1, xs, ys, 0, 1, gg, &v);
        s += l.wscale * v;
    }
    return s;
}

double comp_norm(std::vector<CLev>& L, MultiFab CLev::*fld)
{
    double m = 0.0;
    for (auto& l : L) m = std::max(m, (l.*fld).norm0(0, 1, 0));
    return m;
}

void comp_residual(std::vector<CLev>& L, int lmin = 0)        // r = b - A x on the unknowns of the levels >= lmin
{
    comp_apply(L, lmin, true);
}

}  // namespace

// Projection::initialVelocityProject (Source/Projection.cpp:615-838) on caller-owned arrays: on every level of the solve the pressure is
// zeroed (:640-648), sigma = 1 (rho_wgt_vel_proj = 0) or 1 / rho (:689-730 with scaleVar), rhcc = -divu (:732-743, 783-788), and the
// velocities are projected on the composite grid without touching grad p (increment_gp = false; gp of the levels receives grad phi);
// pres[l] returns phi.  Inflow ghost velocities count in full (inflow_scale 1).
MGStats initial_velocity_project(const std::vector<ProjLevel>& PL, MultiFab* const vel[], const int vcomp[], MultiFab* const pres[], const MultiFab* const rho[],
                                 const int rho_comp[], const MultiFab* const divu[], const int divu_comp[], double proj_tol, double proj_abs_tol, const MGOpts& o)
{
    auto& ctx = Context::get();
    const int nl = (int)PL.size();
    std::vector<MultiFab> sig(nl), rc(nl);
    std::vector<const MultiFab*> sigp(nl), rcp(nl, nullptr);
    bool have_divu = false;
    for (int l = 0; l < nl; ++l) {
        pres[l]->setVal(0.0);
        sig[l].define(PL[l].layout, cell_type(), 1, 0);
        if (rho && rho[l]) {
            const FabD *st = sig[l].d_tab, *rt = rho[l]->d_tab;
            const int rcmp = rho_comp ? rho_comp[l] : 0;
            for_each(*PL[l].layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) { st[f](i, j, k) = 1.0 / rt[f](i, j, k, rcmp); });
        } else sig[l].setVal(1.0);
        sigp[l] = &sig[l];
        if (divu && divu[l]) {
            rc[l].define(PL[l].layout, cell_type(), 1, 0);
            MultiFab::Copy(rc[l], *divu[l], divu_comp ? divu_comp[l] : 0, 0, 1, 0);
            mf_mult(rc[l], -1.0, 0, 1, 0);
            rcp[l] = &rc[l];
            have_divu = true;
        }
    }
    return composite_project(PL, vel, vcomp, pres, sigp.data(), nullptr, proj_tol, proj_abs_tol, false, 1.0, o, have_divu ? rcp.data() : nullptr);
}

// Projection::initialSyncProject (Source/Projection.cpp:970-1185) on caller-owned arrays: U_new <- (U_new - U_old) / dt (ConvertUnew, :1100-1101),
// sigma = 1 / rho_half (scaleVar, :1112-1118), the velocities averaged down level by level (:1120-1136), rhcc = -(divu_new - divu_old) / dt
// (:1008-1075, 1142-1148), composite projection with increment_gp = true (the levels' gp accumulate grad phi); phi[l] (the old-time pressure
// array of the caller, zeroed first, :1002) returns the correction, which is added to pres_new[l] (:1166-1170).  vel_new keeps the projected
// acceleration, as upstream's does until NavierStokesBase::resetState.
MGStats initial_sync_project(const std::vector<ProjLevel>& PL, MultiFab* const vel_new[], const int vcomp[], const MultiFab* const vel_old[], MultiFab* const phi[],
                             MultiFab* const pres_new[], const MultiFab* const rho_half[], const MultiFab* const divu_new[], const MultiFab* const divu_old[],
                             const int divu_comp[], double dt, double proj_tol, double proj_abs_tol, const MGOpts& o)
{
    auto& ctx = Context::get();
    const int nl = (int)PL.size();
    std::vector<MultiFab> sig(nl), rc(nl);
    std::vector<const MultiFab*> sigp(nl), rcp(nl, nullptr);
    bool have_divu = false;
    for (int l = 0; l < nl; ++l) {
        phi[l]->setVal(0.0);
        {
            const double dt_inv = 1. / dt;
            const FabD *nt = vel_new[l]->d_tab, *ot = vel_old[l]->d_tab;
            const int vc = vcomp[l];
            for_each(*PL[l].layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
                for (int n = 0; n < 3; ++n) nt[f](i, j, k, vc + n) = (nt[f](i, j, k, vc + n) - ot[f](i, j, k, vc + n)) * dt_inv;   // ConvertUnew
            });
        }
        sig[l].define(PL[l].layout, cell_type(), 1, 0);
        const FabD *st = sig[l].d_tab, *ht = rho_half[l]->d_tab;
        for_each(*PL[l].layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) { st[f](i, j, k) = 1.0 / ht[f](i, j, k); });
        sigp[l] = &sig[l];
        if (divu_new && divu_new[l] && divu_old && divu_old[l]) {
            const int dc = divu_comp ? divu_comp[l] : 0;
            rc[l].define(PL[l].layout, cell_type(), 1, 0);
            MultiFab::Copy(rc[l], *divu_new[l], dc, 0, 1, 0);
            mf_saxpy(rc[l], -1.0, *divu_old[l], dc, 0, 1, 0);
            mf_mult(rc[l], -1.0 / dt, 0, 1, 0);
            rcp[l] = &rc[l];
            have_divu = true;
        }
    }
    for (int l = nl - 1; l >= 1; --l) {
        MultiFab vf(PL[l].layout, cell_type(), 3, 0), vc(PL[l - 1].layout, cell_type(), 3, 0);
        MultiFab::Copy(vf, *vel_new[l], vcomp[l], 0, 3, 0);
        MultiFab::Copy(vc, *vel_new[l - 1], vcomp[l - 1], 0, 3, 0);
        average_down(vf, vc, 0, 3, PL[l].ratio);
        MultiFab::Copy(*vel_new[l - 1], vc, 0, vcomp[l - 1], 3, 0);
    }
    MGStats st = composite_project(PL, vel_new, vcomp, phi, sigp.data(), nullptr, proj_tol, proj_abs_tol, true, 0.0, o, have_divu ? rcp.data() : nullptr);
    if (pres_new) for (int l = 0; l < nl; ++l) if (pres_new[l]) mf_saxpy(*pres_new[l], 1.0, *phi[l], 0, 0, 1, 1);
    return st;
}

MGStats AmrNS::composite_project(int c0, int nl, MultiFab* const vel[], const int vcomp[], MultiFab* const phi[], const MultiFab* const sig[],
                                 const MultiFab* rhnd, double rtol, double atol, bool increment_gp, double inflow_scale, const MultiFab* const rhcc[])
{
    std::vector<ProjLevel> PL(nl);
    for (int l = 0; l < nl; ++l) PL[l] = proj_level(c0 + l);
    return iamrx::composite_project(PL, vel, vcomp, phi, sig, rhnd, rtol, atol, increment_gp, inflow_scale, o, rhcc);
}

ProjLevel AmrNS::proj_level(int l)
{
    NavierStokes* s = lev[l].get();
    ProjLevel P;
    P.g = s->geom(); P.layout = s->lay(); P.nodal_bc = s->nodal_bc(); P.ratio = s->ratio;
    P.gp = &s->Gp[s->pnew];
    P.set_inflow = [s](MultiFab& v, double scale) { s->set_inflow_ghosts(v, scale); };
    P.fill_gp = [s]() { s->fill_gradp_bc(); };
    return P;
}

MGStats composite_project(const std::vector<ProjLevel>& PL, MultiFab* const vel[], const int vcomp[], MultiFab* const phi[], const MultiFab* const sig[],
                          const MultiFab* rhnd, double rtol, double atol, bool increment_gp, double inflow_scale, const MGOpts& o, const MultiFab* const rhcc[])
{
    auto& ctx = Context::get();
    const int nl = (int)PL.size();
    ProfScope ps_all_("composite_project");
    std::unique_ptr<ProfScope> psec;
    PROF_NEXT(psec, "cp_setup");
    std::vector<CLev> L(nl);
    bool singular = true;
    double scale = 1.0;
    for (int l = 0; l < nl; ++l) {
        CLev& C = L[l];
        C.pl = &PL[l];
        C.g = C.pl->g; C.layout = C.pl->layout;
        C.bc = op_bc(C.pl->nodal_bc);
        const bool has_fine = l < nl - 1;
        for (MultiFab* m : {&C.b, &C.x, &C.r, &C.y, &C.e, &C.bb}) { m->define(C.layout, node_type(), 1, 1); m->setVal(0.0); }
        C.own.define(C.layout, node_type(), 1, 0); C.slave.define(C.layout, node_type(), 1, 0);
        MultiFab cls(C.layout, node_type(), 1, 0);
        MultiFab cov = coverage(C.layout, C.g);
        node_class(cls, cov, C.g);
        MultiFab vsf(C.layout, node_type(), 1, 0);
        vsf.setVal(0.0);
        if (has_fine) { C.fcov = fine_coverage(C.layout, PL[l + 1].layout, C.g, PL[l + 1].ratio); node_class(vsf, C.fcov, C.g); }
        const bool partial = C.layout->total_cells() != C.g.domain.npts();
        bool dir_face = false;
        int dlo[3], dhi[3];
        for (int d = 0; d < 3; ++d) {
            dlo[d] = (!C.g.periodic[d] && C.bc.lo[d] == lo_dirichlet) ? 1 : 0; dhi[d] = (!C.g.periodic[d] && C.bc.hi[d] == lo_dirichlet) ? 1 : 0;
            dir_face = dir_face || dlo[d] || dhi[d];
        }
        if (dir_face || (l == 0 && partial)) singular = false;
        {
            const FabD *ot = C.own.d_tab, *st = C.slave.d_tab, *ct = cls.d_tab, *vt = vsf.d_tab;
            const BoxD dom = C.g.domain;
            const int a0 = dlo[0], a1 = dlo[1], a2 = dlo[2], b0 = dhi[0], b1 = dhi[1], b2 = dhi[2];
            const bool first = l == 0;
            for_each(*C.layout, node_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
                const double c = ct[f](i, j, k);
                double own = c == 1.0 ? 1.0 : 0.0, slave = (c == 2.0 && !first) ? 1.0 : 0.0;
                const bool onD = (a0 && i == dom.lo[0]) || (b0 && i == dom.hi[0] + 1) || (a1 && j == dom.lo[1]) || (b1 && j == dom.hi[1] + 1) || (a2 && k == dom.lo[2]) || (b2 && k == dom.hi[2] + 1);
                if (onD) { own = 0.0; slave = 0.0; }
                if (vt[f](i, j, k) == 1.0) own = 0.0;
                ot[f](i, j, k) = own; st[f](i, j, k) = slave;
            });
        }
        C.wscale = scale;
        scale /= 8.0;
        // sigma on the cells of the level that the next level of the solve does not cover
        C.sigm.define(C.layout, cell_type(), 1, 1);
        C.sigm.setVal(0.0);
        MultiFab::Copy(C.sigm, *sig[l], 0, 0, 1, 0);
        if (has_fine) mask_mult(C.sigm, 0, 1, C.fcov, true, 0);
        C.sigm.FillBoundary(C.g);
        cc_mirror_bc(C.g, C.sigm);
        MGOpts mo = o;
        // IAMRX_CP_FINE_SMOOTH = K > 0: the correction of a REFINED level of the solve is K smoother calls on the level itself -- no hierarchy
        // below it, no bottom solve: the level underneath is its coarse grid, as in amrex::MLMG's composite cycle (mgFcycle / oneIter: smooth,
        // restrict the residual to the AMR level below, interpolate its correction, smooth).  0: a V-cycle of the level's own hierarchy.
        const int fine_smooth = (int)tune("CP_FINE_SMOOTH", 1);
        if (l > 0 && fine_smooth > 0) {
            mo.max_coarsening_level = 0; mo.bottom_smoother_only = 1; mo.nuf = fine_smooth;
            const int sw = (int)tune("CP_FINE_SWEEPS", 0);             // > 0: sweeps per smoother call of these corrections (default: the solver's)
            if (sw > 0) mo.nodal_sweeps = sw;
        }
        C.mg = std::make_unique<NodalMG>(C.g, C.layout, C.pl->nodal_bc, mo);
        C.mg->setSigma(*sig[l], 0);
    }
    PROF_NEXT(psec, "cp_rhs");
    // right-hand side: div(vel) of the uncovered cells (+ rhnd), fine boundary contributions handed down
    for (int l = nl - 1; l >= 0; --l) {
        CLev& C = L[l];
        vel[l]->FillBoundary(C.g, vcomp[l], 3);
        if (C.pl->set_inflow) C.pl->set_inflow(*vel[l], inflow_scale);
        MultiFab um(C.layout, cell_type(), 3, 1);
        um.setVal(0.0);
        MultiFab::Copy(um, *vel[l], vcomp[l], 0, 3, 0);
        if (l < nl - 1) mask_mult(um, 0, 3, C.fcov, true, 0);
        um.FillBoundary(C.g);
        {
            const FabD *ut = um.d_tab, *vt = vel[l]->d_tab;
            const BoxD dom = C.g.domain;
            const int p0 = C.g.periodic[0], p1 = C.g.periodic[1], p2 = C.g.periodic[2], vc = vcomp[l];
            for_each(*C.layout, cell_type(), 1, ctx.stream, [=] __device__(int i, int j, int k, int f) {
                const bool out = (!p0 && (i < dom.lo[0] || i > dom.hi[0])) || (!p1 && (j < dom.lo[1] || j > dom.hi[1])) || (!p2 && (k < dom.lo[2] || k > dom.hi[2]));
                if (out) for (int n = 0; n < 3; ++n) ut[f](i, j, k, n) = vt[f](i, j, k, vc + n);
            });
        }
        MultiFab dv(C.layout, node_type(), 1, 0);
        const DomainBC bcn = C.pl->nodal_bc;
        nodal_divu(C.g, dv, um, 0, &bcn);
        if (rhcc && rhcc[l]) { MultiFab rc = make_rhcc(C.g, *rhcc[l], 0, 1.0, l < nl - 1 ? &C.fcov : nullptr); nodal_rhcc_add(C.g, dv, rc, bcn); }
        mf_saxpy(C.b, 1.0, dv, 0, 0, 1, 0);                    // b may already hold what the finer level handed down
        if (l == 0 && rhnd) mf_saxpy(C.b, 1.0, *rhnd, 0, 0, 1, 0);
        if (l > 0) {
            MultiFab bb(C.layout, node_type(), 1, 1);
            bb.setVal(0.0);
            MultiFab::Copy(bb, C.b, 0, 0, 1, 0);
            mask_mult(bb, 0, 1, C.slave, false, 0);
            restrict_to_crse(L[l - 1].b, bb, C.g, L[l - 1].g, C.pl->ratio);
        }
        mask_mult(C.b, 0, 1, C.own, false, 0);
    }
    for (int l = 0; l < nl; ++l) MultiFab::Copy(L[l].x, *phi[l], 0, 0, 1, 0);
    if (singular) {                      // MLMG::makeSolvable: remove the mean of the right-hand side over the composite unknowns
        const double off = comp_dot_own(L, 0) / comp_dot_own(L, 1);
        for (auto& C : L) { mf_add_scalar(C.b, -off, 0, 1, 0); mask_mult(C.b, 0, 1, C.own, false, 0); }
    }
    PROF_NEXT(psec, "cp_cycles");
    MGStats st;
    st.nlevels = nl;
    comp_residual(L);
    st.rhsnorm0 = comp_norm(L, &CLev::b);
    st.resnorm0 = comp_norm(L, &CLev::r);
    st.resnorm = st.resnorm0;
    const double max_norm = std::max(st.rhsnorm0, st.resnorm0);
    const double target = std::max(atol, std::max(rtol, 1.e-16) * max_norm);
    if (st.resnorm0 <= target) st.converged = 1;
    MGStats vst;
    // one subspace correction: the space of level l = its trilinear functions vanishing on the level's boundary, prolonged to the
    // unknowns of every finer level.  Right-hand side = the composite residual tested with those functions: the level's own entries
    // plus the full-weighting restriction of the finer levels' entries.
    int fresh_from = 0;                  // L[m].r is current for m >= fresh_from (nl: nothing is)
    auto correct_level = [&](int l) {
        if (fresh_from > l) { ProfScope ps("cp_residual"); comp_residual(L, l); }
        fresh_from = nl;                 // the correction below changes x
        ProfScope* ps_dn = new ProfScope("cp_restrict_rhs");
        MultiFab acc;                                          // on level m-1: what levels >= m hand down
        auto r_plus = [](MultiFab& out, const MultiFab& r, const MultiFab* add) {      // out = r (+ add) on the valid nodes (ghost nodes stay zero)
            const FabD *ot = out.d_tab, *rt = r.d_tab, *at = add ? add->d_tab : nullptr;
            for_each(*out.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
                ot[f](i, j, k) = at ? rt[f](i, j, k) + 1.0 * at[f](i, j, k) : (double)rt[f](i, j, k);
            });
        };
        for (int m = nl - 1; m > l; --m) {
            r_plus(L[m].bb, L[m].r, m < nl - 1 ? &acc : nullptr);
            MultiFab down(L[m - 1].layout, node_type(), 1, 0);
            restrict_to_crse(down, L[m].bb, L[m].g, L[m - 1].g, L[m].pl->ratio);
            acc = std::move(down);
        }
        // the right-hand side goes straight into the level solver's residual array and the correction is read from its correction array
        // (valid nodes + the ghost layer it fills): no copies in and out of the V-cycle
        r_plus(L[l].mg->res(0), L[l].r, l < nl - 1 ? &acc : nullptr);
        delete ps_dn;
        { ProfScope ps("cp_vcycle"); L[l].mg->vcycle_correction_inplace(vst); }
        ProfScope ps_up("cp_interp_update");
        MultiFab& el = L[l].mg->cor(0);
        for (int m = l; m < nl; ++m) {
            MultiFab& em = m == l ? el : L[m].e;
            if (m > l) {                                        // e of level m = the interpolant of the coarser correction (all nodes)
                MultiFab& ec = m - 1 == l ? el : L[m - 1].e;
                if (m - 1 > l) fill_nodes(L[m - 1], ec);        // (the level solver has filled the ghost nodes of its own correction)
                if (m == nl - 1) {                              // nobody interpolates from the finest level: x += interpolant on its unknowns, one pass
                    node_interp_from_crse(L[m].x, ec, L[m - 1].g, L[m].pl->ratio, &L[m].own, true);
                    continue;
                }
                node_interp_from_crse(em, ec, L[m - 1].g, L[m].pl->ratio, nullptr, false);
            }
            const FabD *xt = L[m].x.d_tab, *et = em.d_tab, *ot = L[m].own.d_tab;                 // x += e on the unknowns
            for_each(*L[m].layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
                if (ot[f](i, j, k) != 0.0) xt[f](i, j, k) += 1.0 * et[f](i, j, k);
            });
        }
    };
    std::vector<double> hist;
    for (int it = 0; it < o.max_iters && !st.converged; ++it) {
        // symmetric sweep: finest first, down to the coarsest level of the solve, and up again.  From the second iteration on the downward
        // half starts one level below the finest: the upward half of the iteration before has just ended with a correction of the finest
        // level and nothing has changed since (two corrections of the same space in a row: the second finds only what the first left)
        for (int l = nl - 1 - (it > 0 && nl > 1 ? 1 : 0); l >= 1; --l) correct_level(l);
        correct_level(0);
        for (int l = 1; l < nl; ++l) correct_level(l);
        { ProfScope ps("cp_residual"); comp_residual(L); }
        fresh_from = 0;
        st.resnorm = comp_norm(L, &CLev::r);
        st.iters = it + 1;
        if (o.verbose) printf("iamrx composite nodal solve: iter %d resid %.6e (target %.3e)\n", it + 1, st.resnorm, target);
        if (st.resnorm <= target) { st.converged = 1; break; }
        if (!(st.resnorm < 1.e20 * max_norm)) throw Error("iamrx composite nodal solve: residual blow-up");
        // the round-off floor of the fp64 residual (about 1e-12 of the right-hand side once h <= 1/512) can sit just above the requested
        // tolerance: a residual within 10x of the target that has lost less than 10 % over three iterations is as converged as fp64 allows.
        // amrex::MLMG would iterate to max_iter and abort; converged = 2 reports the difference
        hist.push_back(st.resnorm);
        if (mg_stalled_at_floor(hist, st.resnorm0, target, tune("MG_STALL_FAST", 1) != 0)) {
            st.converged = 2;
            fprintf(stderr, "iamrx composite nodal solve: WARNING: residual %.3e stalled at the round-off floor above the target %.3e after %d iterations; accepted (converged = 2)\n",
                    st.resnorm, target, st.iters);
            break;
        }
    }
    if (!st.converged) throw Error("iamrx composite nodal solve: failed to converge");
    if (singular) {                      // the solution of the singular system is fixed by a zero weighted mean over the composite unknowns
        const double off = comp_dot_own(L, 2) / comp_dot_own(L, 1);
        for (auto& C : L) {
            MultiFab t(C.layout, node_type(), 1, 0);
            MultiFab::Copy(t, C.own, 0, 0, 1, 0);
            mf_saxpy(C.x, -off, t, 0, 0, 1, 0);
        }
    }
    PROF_NEXT(psec, "cp_finish");
    // slaves, covered coarse nodes (injection of the fine solution), ghost nodes
    comp_fill_slaves(L);
    for (int l = nl - 2; l >= 0; --l) { average_down(L[l + 1].x, L[l].x, 0, 1, L[l + 1].pl->ratio); fill_nodes(L[l], L[l].x); }
    for (int l = 0; l < nl; ++l) {
        CLev& C = L[l];
        MultiFab::Copy(*phi[l], C.x, 0, 0, 1, 1);
        nodal_mknewu(C.g, vel[l], vcomp[l], *phi[l], sig[l], C.pl->gp, increment_gp);       // (sigma is component 0 of sig[l], as for setSigma above)
    }
    for (int l = nl - 1; l >= 1; --l) {                  // NodalProjector::averageDown(vel)
        if (vcomp[l] == vcomp[l - 1]) { average_down(*vel[l], *vel[l - 1], vcomp[l], 3, L[l].pl->ratio); continue; }
        MultiFab vf(L[l].layout, cell_type(), 3, 0), vc(L[l - 1].layout, cell_type(), 3, 0);
        MultiFab::Copy(vf, *vel[l], vcomp[l], 0, 3, 0);
        MultiFab::Copy(vc, *vel[l - 1], vcomp[l - 1], 0, 3, 0);
        average_down(vf, vc, 0, 3, L[l].pl->ratio);
        MultiFab::Copy(*vel[l - 1], vc, 0, vcomp[l - 1], 3, 0);
    }
    for (int l = 0; l < nl; ++l) if (L[l].pl->fill_gp) L[l].pl->fill_gp();
    return st;
}

// ------------------------------------------------------------------------------------------------------------------------
AmrNS::AmrNS(const Geometry& g0, const std::vector<LayoutP>& layouts, int ratio, const NSParams& p_, const MGOpts& o_) : p(p_), o(o_)
{
    IAMRX_ASSERT(ratio == 2 && !layouts.empty());
    Geometry g = g0;
    for (size_t l = 0; l < layouts.size(); ++l) {
        if (l > 0) {
            for (int d = 0; d < 3; ++d) { g.domain.lo[d] *= ratio; g.domain.hi[d] = (g.domain.hi[d] + 1) * ratio - 1; g.dx[d] /= (double)ratio; }
        }
        lev.push_back(std::make_unique<NavierStokes>(g, layouts[l], p, o));
        NavierStokes& s = *lev.back();
        s.level = (int)l; s.ratio = l > 0 ? ratio : 1;
        n_cycle.push_back(l > 0 ? ratio : 1);
        dt_level.push_back(0.0); dt_min.push_back(1.e200);
        if (l > 0) {
            NavierStokes& c = *lev[l - 1];
            if (l > 1) check_nesting(layouts[l]->boxes, layouts[l - 1]->boxes, c.g, (int)l);
            link_level((int)l);
        }
    }
}

// proper nesting: the Godunov ghost cells (3) of every box plus the interpolation stencil (1 coarse cell) must lie inside the next
// coarser level or outside a non-periodic domain face -- what amrex::Amr's grid generation guarantees for IAMR (blocking_factor >= 8).
// Layouts that violate it would read cells no level defines.
void AmrNS::check_nesting(const std::vector<BoxD>& fine, const std::vector<BoxD>& crse, const Geometry& cgeom, int l) const
{
    const int ratio = m_ratio;
    const BoxD cdom = cgeom.domain;
    for (auto& fb : fine) {
        BoxD R = grow(coarsen(grow(fb, 3), ratio), 1);
        for (int d = 0; d < 3; ++d) if (!cgeom.periodic[d]) { R.lo[d] = std::max(R.lo[d], cdom.lo[d]); R.hi[d] = std::min(R.hi[d], cdom.hi[d]); }
        long covered = 0;
        for (auto& cb : crse)
            for (int sz = -1; sz <= 1; ++sz) for (int sy = -1; sy <= 1; ++sy) for (int sx = -1; sx <= 1; ++sx) {
                const int sh[3] = {sx, sy, sz};
                bool ok = true;
                BoxD q = cb;
                for (int d = 0; d < 3; ++d) {
                    if (sh[d] != 0 && !cgeom.periodic[d]) ok = false;
                    q.lo[d] += sh[d] * cdom.len(d); q.hi[d] += sh[d] * cdom.len(d);
                }
                if (!ok) continue;
                const BoxD is = intersect(R, q);
                if (is.ok()) covered += is.npts();
            }
        if (covered != R.npts()) throw Error("iamrx AmrNS: level " + std::to_string(l) + " is not properly nested in level " + std::to_string(l - 1) +
                                             " (3 ghost cells + 1 coarse stencil cell must lie inside the coarser level)");
    }
}

// the interface data of level l > 0 and its coarser level: registers (owned by the fine level), rho_avg / p_avg, Vsync / Ssync of the coarse one
void AmrNS::link_level(int l)
{
    NavierStokes &s = *lev[l], &c = *lev[l - 1];
    const int ratio = s.ratio;
    s.crse = &c; c.fine = &s;
    s.rho_avg.define(s.layout, cell_type(), 1, 1); s.rho_avg.setVal(0.0);
    s.p_avg.define(s.layout, node_type(), 1, 0); s.p_avg.setVal(0.0);
    s.reg_adv = std::make_unique<FluxRegister>(s.layout, c.layout, c.g, ratio, s.nstate);
    s.reg_visc = std::make_unique<FluxRegister>(s.layout, c.layout, c.g, ratio, s.nstate);
    s.reg_mac = std::make_unique<FluxRegister>(s.layout, c.layout, c.g, ratio, 1);
    s.sync_reg = std::make_unique<SyncRegister>(s.layout, c.layout, c.g, s.g, ratio, p.phys_lo, p.phys_hi);
    c.Vsync.define(c.layout, cell_type(), 3, 1); c.Vsync.setVal(0.0);
    c.Ssync.define(c.layout, cell_type(), c.nstate - 3, 1); c.Ssync.setVal(0.0);
}

namespace {
struct AmrTimer {
    AmrNS& a; int idx; bool on;
    std::chrono::steady_clock::time_point t0;
    AmrTimer(AmrNS& a_, int i) : a(a_), idx(i), on(a_.profile_on) { if (on) { Context::get().sync(); t0 = std::chrono::steady_clock::now(); } }
    ~AmrTimer() { if (on) { Context::get().sync(); a.t_prof[idx] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } }
};
}  // namespace

// NavierStokes::avgDown (NavierStokes.cpp:1845-1873) + avgDown_StatePress (NavierStokesBase.cpp:4125-4163)
void AmrNS::avg_down(int l)
{
    NavierStokes &c = *lev[l], &f = *lev[l + 1];
    average_down(f.S[f.inew], c.S[c.inew], 0, c.nalloc, f.ratio);     // state, and divu / dsdt (NavierStokes.cpp:1859-1872)
    for (size_t q = l; q < lev.size(); ++q) lev[q]->make_rho_curr_time();
    average_down(c.initial_step ? f.P[f.pnew] : f.p_avg, c.P[c.pnew], 0, 1, f.ratio);
    average_down(f.Gp[f.pnew], c.Gp[c.pnew], 0, 3, f.ratio);
    // The reference leaves the ghost cells of the coarse Gradp as they were (filled before the average), which makes the next
    // predictor depend on how the coarse level happens to be chopped into boxes.  Here they are re-filled.
    c.fill_gradp_bc();
}

// NavierStokes::reflux (NavierStokes.cpp:1736-1838)
void AmrNS::reflux(int l)
{
    NavierStokes &c = *lev[l], &f = *lev[l + 1];
    auto& ctx = Context::get();
    const double vol = c.g.dx[0] * c.g.dx[1] * c.g.dx[2], dt_crse = dt_level[l];
    f.reg_visc->Reflux(c.Vsync, vol, 1.0, 0, 0, 3);
    f.reg_visc->Reflux(c.Ssync, vol, 1.0, 3, 0, c.nstate - 3);
    {
        const FabD *vt = c.Vsync.d_tab, *st = c.Ssync.d_tab, *ht = c.rho_half.d_tab;
        const bool mom = c.p.do_mom_diff != 0;
        ScalForm sf;
        for (int n = 0; n < MAXSCAL; ++n) sf.form[n] = c.scal_cons[n];
        const int ns_ = c.nscal;
        for_each(*c.layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int fb) {
            const double rh = ht[fb](i, j, k);
            if (!mom) for (int n = 0; n < 3; ++n) vt[fb](i, j, k, n) /= rh;
            for (int n = 1; n < ns_; ++n) if (!sf.form[n]) st[fb](i, j, k, n) /= rh;   // the non-conservative scalars (density is conservative)
        });
    }
    f.reg_adv->Reflux(c.Vsync, vol, 1.0, 0, 0, 3);
    f.reg_adv->Reflux(c.Ssync, vol, 1.0, 3, 0, c.nstate - 3);
    mf_mult(c.Vsync, 1.0 / dt_crse, 0, 3, 1);
    mf_mult(c.Ssync, 1.0 / dt_crse, 0, c.nstate - 3, 1);
    // zero the coarse cells under the fine grids (grown tile box: ghost cells included)
    MultiFab fc = fine_coverage(c.layout, f.layout, c.g, f.ratio);
    mask_mult(c.Vsync, 0, 3, fc, true, 1);
    mask_mult(c.Ssync, 0, c.nstate - 3, fc, true, 1);
}

// NavierStokes::mac_sync (NavierStokes.cpp:1438-1730) with MacProj::mac_sync_solve / mac_sync_compute; non-diffusive scalars,
// inviscid velocity
void AmrNS::mac_sync(int l)
{
    NavierStokes &c = *lev[l], &f = *lev[l + 1];
    auto& ctx = Context::get();
    const double dt = dt_level[l];
    MultiFab Ucorr[3];
    MultiFab* uc[3];
    for (int d = 0; d < 3; ++d) { Ucorr[d].define(c.layout, face_type(d), 1, 1); Ucorr[d].setVal(0.0); uc[d] = &Ucorr[d]; }
    MGOpts mo = o;
    mo.maxorder = 4;
    {
        AmrTimer t(*this, 2);
        st_mac_sync = mac_sync_solve(c.g, *f.reg_mac, c.rho_half, dt, f.layout, f.ratio, uc, c.mac_phi, c.bc_mac, 1.e-10 /*mac_sync_tol*/, c.p.mac_abs_tol, mo,
                                     l > 0 ? &c.crse->g : nullptr, c.ratio);
    }
    AmrTimer t_rest(*this, 3);
    ProfScope ps_ms_("mac_sync_rest");
    std::unique_ptr<ProfScope> psec;
    PROF_NEXT(psec, "ms_compute");
    for (int d = 0; d < 3; ++d) Ucorr[d].FillBoundary(c.g);
    // ---- mac_sync_compute (MacProj.cpp:490-731)
    {
        const bool mom = c.p.do_mom_diff != 0;
        MultiFab Smf(c.layout, cell_type(), 3, 3), Sc(c.layout, cell_type(), c.nscal, 3);
        c.fillpatch(Smf, c.S[1 - c.inew], Xvel, 3, c.bc_vel);
        c.fillpatch(Sc, c.S[1 - c.inew], Density, c.nscal, c.bc_scal);
        if (mom) {
            const FabD *ut = Smf.d_tab, *rt = Sc.d_tab;
            for_each(*c.layout, cell_type(), 3, ctx.stream, [=] __device__(int i, int j, int k, int fb) {
                const double r = rt[fb](i, j, k, 0);
                for (int n = 0; n < 3; ++n) ut[fb](i, j, k, n) *= r;
            });
        }
        MultiFab tfs(c.layout, cell_type(), c.nscal, 1), divu;
        tfs.setVal(0.0);
        c.divu_half(divu, dt, 1, false);                             // getDivCond(nghost_force, prev_time), MacProj.cpp:562
        // viscous forcing at the old time (MacProj.cpp:566-572): getViscTerms(visc_terms, 0, num_state_comps, prev_time)
        MultiFab vvisc(c.layout, cell_type(), 3, 1);
        vvisc.setVal(0.0);
        if (c.p.be_cn_theta != 1.0) {
            if (c.is_diffusive_vel()) c.get_visc_terms_vel(vvisc, c.S[1 - c.inew]);
            for (int n = 1; n < c.nscal; ++n) {
                if (!c.is_diffusive_scal(Density + n)) continue;
                // conservative scalar: tf += visc, convective: tf = tf / rho + visc with tf = 0, temperature: (tf + visc) / rho (MacProj.cpp:641-683)
                MultiFab sv(c.layout, cell_type(), 1, 1);
                c.get_visc_terms_scalar(sv, c.S[1 - c.inew], Density + n);
                if (Density + n == c.Temp) scale_by(sv, Sc, 0, 1, true);
                MultiFab::Copy(tfs, sv, 0, n, 1, 1);
            }
        }
        MultiFab* um[3] = {&c.u_mac[0], &c.u_mac[1], &c.u_mac[2]};
        MultiFab flv[3], fls[3];
        MultiFab *flvp[3], *flsp[3];
        for (int d = 0; d < 3; ++d) { flv[d].define(c.layout, face_type(d), 3, 0); fls[d].define(c.layout, face_type(d), c.nscal, 0); flvp[d] = &flv[d]; flsp[d] = &fls[d]; }
        mac_sync_compute(c.g, uc, c.Vsync, c.Ssync, Smf, Sc, c.nscal, &vvisc, &tfs, c.Gp[1 - c.pnew], &divu, um, c.scal_cons, mom, c.p.gravity, dt, c.bc_vel,
                         c.bc_scal, c.p.use_forces_in_trans != 0, c.p.use_ppm, flvp, flsp);
        for (int d = 0; d < 3; ++d) {            // NavierStokesBase.cpp:5083-5096 with sync_factor = -1
            f.reg_adv->CrseInit(flv[d], d, 0, 0, 3, dt, true);
            f.reg_adv->CrseInit(fls[d], d, 0, Density, c.nscal, dt, true);
            if (l > 0) {                         // this level is itself the fine side of the interface below
                c.reg_adv->FineAdd(flv[d], d, 0, 0, 3, -dt);
                c.reg_adv->FineAdd(fls[d], d, 0, Density, c.nscal, -dt);
                c.reg_mac->FineAdd(Ucorr[d], d, 0, 0, 1, -c.g.dx[(d + 1) % 3] * c.g.dx[(d + 2) % 3] / (double)n_cycle[l]);   // MacProj.cpp:720-727
            }
        }
    }
    PROF_NEXT(psec, "ms_update");
    // ---- NavierStokes.cpp:1490-1690
    MultiFab& Sn = c.S[c.inew];
    MultiFab Delta(c.layout, cell_type(), c.nscal, 0);
    const bool mom = c.p.do_mom_diff != 0;
    {
        const FabD *st = c.Ssync.d_tab, *vt = c.Vsync.d_tab, *nt = Sn.d_tab, *dt_ = Delta.d_tab;
        ScalForm sf;
        for (int n = 0; n < MAXSCAL; ++n) sf.form[n] = c.scal_cons[n];
        const int ns_ = c.nscal;
        for_each(*c.layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int fb) {
            const double rho = nt[fb](i, j, k, Density);
            for (int n = 1; n < ns_; ++n) if (sf.form[n]) {   // conservative Q = rho q: sync -= (sync of rho) * q, added back below
                const double dl = nt[fb](i, j, k, Density + n) * st[fb](i, j, k, 0) / rho;
                dt_[fb](i, j, k, n) = dl;
                st[fb](i, j, k, n) -= dl;
            }
            if (mom) for (int n = 0; n < 3; ++n) vt[fb](i, j, k, n) /= rho;
        });
    }
    const double theta = c.p.be_cn_theta;
    PROF_NEXT(psec, "ms_vsync_diffuse");
    if (c.is_diffusive_vel()) {
        // Diffusion::diffuse_Vsync -> diffuse_tensor_Vsync (Diffusion.cpp:960-1178): (rho - theta dt div tau) Vsync' = rho Vsync, homogeneous
        // boundary and coarse/fine data.  NOTE the face coefficients of this solve are set to 1.0 upstream (:1122-1135), not to the
        // viscosity -- followed as written.
        MultiFab one[3];
        const MultiFab* ep[3];
        for (int d = 0; d < 3; ++d) { one[d].define(c.layout, face_type(d), 1, 0); one[d].setVal(1.0); ep[d] = &one[d]; }
        MultiFab tflux[3];
        MultiFab* tfp[3] = {&tflux[0], &tflux[1], &tflux[2]};
        if (l > 0) for (int d = 0; d < 3; ++d) tflux[d].define(c.layout, face_type(d), 3, 0);
        diffuse_tensor_Vsync(c.g, c.Vsync, dt, theta, c.rho_half, mom ? 3 : 1 /* NavierStokes.cpp:1552 */, &c.S[1 - c.inew], &Sn, Density, ep, c.bc_visc, c.bc_vel,
                             l > 0 ? &c.crse->g : nullptr, c.ratio, l > 0 ? tfp : nullptr, c.p.visc_tol, o);
        if (l > 0) for (int d = 0; d < 3; ++d) c.reg_visc->FineAdd(tflux[d], d, 0, Xvel, 3, dt * dt);     // :1166-1176
    }
    PROF_NEXT(psec, "ms_ssync");
    mf_mult(c.Ssync, dt, 0, 1, 1);                           // density: not diffusive: Ssync.mult(dt, sigma, 1, ngrow)
    for (int sn = 1; sn < c.nscal; ++sn) {
    const int sigma = Density + sn, rho_flag = c.scal_rho_flag[sn];
    if (c.is_diffusive_scal(sigma)) {
        // Diffusion::diffuse_scalar as the sync solve (NavierStokes.cpp:1590-1640), fluxSC -> viscous register x dt (:1630-1638)
        const MultiFab* bp[3] = {&c.diff_b[sn][0], &c.diff_b[sn][1], &c.diff_b[sn][2]};
        MultiFab sf[3];
        MultiFab* sfp[3] = {&sf[0], &sf[1], &sf[2]};
        if (l > 0) for (int d = 0; d < 3; ++d) sf[d].define(c.layout, face_type(d), 1, 0);
        diffuse_Ssync(c.g, c.Ssync, sn, dt, theta, c.rho_half, rho_flag, Sn, Density, bp, c.bc_scal_lin[sn], l > 0 ? &c.crse->g : nullptr, c.ratio,
                      l > 0 ? sfp : nullptr, c.p.visc_tol, o);
        if (l > 0) for (int d = 0; d < 3; ++d) c.reg_visc->FineAdd(sf[d], d, 0, sigma, 1, dt);
    } else mf_mult(c.Ssync, dt, sn, 1, 1);
    if (c.scal_cons[sn]) mf_saxpy(c.Ssync, dt, Delta, sn, sn, 1, 0);
    }
    mf_saxpy(Sn, 1.0, c.Ssync, 0, Density, c.nstate - 3, 0);
    c.make_rho_curr_time();
    if (l > 0) mf_saxpy(c.rho_avg, 1.0, c.Ssync, 0, 0, 1, 0);      // :1684-1688
    PROF_NEXT(psec, "ms_sync_interp");
    // interpolate the sync correction to every finer level, straight from this one with the accumulated ratio (:1697-1725)
    int ratio = 1;
    for (size_t q = (size_t)l + 1; q < lev.size(); ++q) {
        NavierStokes& ff = *lev[q];
        ratio *= ff.ratio;
        MultiFab incr(ff.layout, cell_type(), c.nstate - 3, 0);
        sync_interp_cellcons(incr, 0, c.Ssync, 0, c.nstate - 3, c.g, ff.g, ratio, c.bc_scal);
        mf_saxpy(ff.S[ff.inew], 1.0, incr, 0, Density, c.nstate - 3, 0);
        ff.make_rho_curr_time();
        mf_saxpy(ff.rho_avg, 1.0, incr, 0, 0, 1, 0);
    }
}

// Projection::MLsyncProject (Projection.cpp:457-607) on caller-owned data, see operators.h
MGStats ml_sync_project(const ProjLevel PL[2], MultiFab& pres_crse, MultiFab& vel_crse, int vcomp_c, MultiFab& pres_fine, MultiFab& vel_fine, int vcomp_f,
                        const MultiFab& rho_crse, const MultiFab& rho_fine, MultiFab& Vsync, MultiFab& V_corr, MultiFab& phi_c, MultiFab& phi_f,
                        SyncRegister& rhs_sync_reg, SyncRegister* crse_sync_reg, double dt, int crse_iteration, int crse_dt_ratio,
                        double sync_tol, double abs_tol, const MGOpts& o)
{
    auto& ctx = Context::get();
    const ProjLevel &C = PL[0], &F = PL[1];
    MultiFab rhnd(C.layout, node_type(), 1, 0);
    phi_c.setVal(0.0); phi_f.setVal(0.0);
    rhs_sync_reg.InitRHS(rhnd);
    if (F.layout->boxes.size() == 1 && F.layout->boxes[0].npts() == F.g.domain.npts()) rhnd.setVal(0.0);   // Projection.cpp:506-510
    // scaleVar: sigma = 1/rho (rho_half on the coarse level, rho_avg on the fine one); then velocity and sigma averaged down
    MultiFab sig_c(C.layout, cell_type(), 1, 0), sig_f(F.layout, cell_type(), 1, 0);
    {
        const FabD *st = sig_c.d_tab, *ht = rho_crse.d_tab;
        for_each(*C.layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int fb) { st[fb](i, j, k) = 1.0 / ht[fb](i, j, k); });
        const FabD *sf = sig_f.d_tab, *hf = rho_fine.d_tab;
        for_each(*F.layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int fb) { sf[fb](i, j, k) = 1.0 / hf[fb](i, j, k); });
    }
    {
        MultiFab vf(F.layout, cell_type(), 3, 0), vc(C.layout, cell_type(), 3, 0);
        MultiFab::Copy(vf, V_corr, 0, 0, 3, 0);
        MultiFab::Copy(vc, Vsync, 0, 0, 3, 0);
        average_down(vf, vc, 0, 3, F.ratio);
        MultiFab::Copy(Vsync, vc, 0, 0, 3, 0);
        average_down(sig_f, sig_c, 0, 1, F.ratio);
    }
    MultiFab* vel[2] = {&Vsync, &V_corr};
    const int vcomp[2] = {0, 0};
    MultiFab* phi[2] = {&phi_c, &phi_f};
    const MultiFab* sig[2] = {&sig_c, &sig_f};
    // Projection.cpp:544-569: a sync projection above levels 0-1 changes the level-l velocity; the residual of the composite solution
    // on the boundary nodes of level l goes to the sync register of the interface below (SyncRegister::CompAdd, SyncRegister.cpp:321-348)
    const bool want_resid = crse_sync_reg != nullptr && crse_iteration == crse_dt_ratio;
    MultiFab vold_c;
    if (want_resid) {
        Vsync.FillBoundary(C.g);
        vold_c.define(C.layout, cell_type(), 3, 1);
        MultiFab::Copy(vold_c, Vsync, 0, 0, 3, 1);
    }
    std::vector<ProjLevel> pl = {C, F};
    MGStats st = composite_project(pl, vel, vcomp, phi, sig, &rhnd, sync_tol, abs_tol, true, 0.0, o);
    if (want_resid) {
        MultiFab r = sync_resid(C.g, C.layout, C.nodal_bc, LayoutP(), 2, vold_c, phi_c, sig_c);
        crse_sync_reg->CompAdd(r, C.g, F.layout, F.ratio, 1.0 / (double)crse_dt_ratio);
    }
    mf_saxpy(pres_crse, 1.0, phi_c, 0, 0, 1, 1);
    mf_saxpy(pres_fine, 1.0, phi_f, 0, 0, 1, 1);
    mf_saxpy(vel_crse, dt, Vsync, 0, vcomp_c, 3, 1);
    mf_saxpy(vel_fine, dt, V_corr, 0, vcomp_f, 3, 1);
    return st;
}

// NavierStokesBase::level_sync (NavierStokesBase.cpp:1927-2044) + Projection::MLsyncProject (Projection.cpp:457-607)
void AmrNS::level_sync(int l, int crse_iteration)
{
    NavierStokes &c = *lev[l], &f = *lev[l + 1];
    const int crse_dt_ratio = n_cycle[l];
    if (crse_iteration < 0) crse_iteration = crse_dt_ratio;
    const double dt = dt_level[l];
    ProfScope ps_ls_("level_sync");
    std::unique_ptr<ProfScope> psec;
    PROF_NEXT(psec, "ls_pre");
    c.Vsync.FillBoundary(c.g);
    MultiFab V_corr(f.layout, cell_type(), 3, 1);
    V_corr.setVal(0.0);
    sync_interp_cellcons(V_corr, 0, c.Vsync, 0, 3, c.g, f.g, f.ratio, c.bc_vel);          // SyncInterp, increment = 0
    MultiFab phi_c(c.layout, node_type(), 1, 1), phi_f(f.layout, node_type(), 1, 1);
    psec.reset();
    const ProjLevel PL[2] = {proj_level(l), proj_level(l + 1)};
    st_sync = ml_sync_project(PL, c.P[c.pnew], c.S[c.inew], Xvel, f.P[f.pnew], f.S[f.inew], Xvel, c.rho_half, f.rho_avg, c.Vsync, V_corr, phi_c, phi_f,
                              *f.sync_reg, l > 0 ? c.sync_reg.get() : nullptr, dt, crse_iteration, crse_dt_ratio, 1.e-10 /*sync_tol*/, c.p.proj_abs_tol, o);
    PROF_NEXT(psec, "ls_post");
    // NavierStokesBase.cpp:2018-2040: the levels above l+1 get the interpolated velocity correction (SyncInterp, increment, x dt) and
    // pressure correction (SyncProjInterp: node_bilinear_interp of phi, added to P_new AND P_old), then computeGradP at both times
    int ratio = 1;
    for (size_t q = (size_t)l + 2; q < lev.size(); ++q) {
        NavierStokes& ff = *lev[q];
        ratio *= ff.ratio;
        MultiFab Vi(ff.layout, cell_type(), 3, 0);
        sync_interp_cellcons(Vi, 0, V_corr, 0, 3, f.g, ff.g, ratio, f.bc_vel);
        mf_saxpy(ff.S[ff.inew], dt, Vi, 0, Xvel, 3, 0);
        MultiFab pi(ff.layout, node_type(), 1, 0);
        pi.setVal(0.0);
        node_interp_from_crse(pi, phi_f, f.g, ratio, nullptr, false);
        for (int w = 0; w < 2; ++w) {
            mf_saxpy(ff.P[w], 1.0, pi, 0, 0, 1, 0);
            nodal_mknewu(ff.g, nullptr, 0, ff.P[w], nullptr, &ff.Gp[w], false);          // computeGradP
            ff.fill_gp(ff.Gp[w], w == ff.pnew ? 0.5 * (ff.pt_new[0] + ff.pt_new[1]) : 0.5 * (ff.pt_old[0] + ff.pt_old[1]));
        }
    }
}


void AmrNS::set_profile(bool on)
{
    profile_on = on;
    for (auto& s : lev) s->profile_sections = on;
    if (on) { for (double& t : t_prof) t = 0.0; for (auto& s : lev) for (double& t : s->t_sections) t = 0.0; }
}

// NavierStokesBase::post_timestep (NavierStokesBase.cpp:2546-2636)
void AmrNS::post_timestep(int l, int crse_iteration)
{
    if (l < (int)lev.size() - 1) {
        { AmrTimer t(*this, 0); reflux(l); }
        { AmrTimer t(*this, 1); avg_down(l); }
        mac_sync(l);
        { AmrTimer t(*this, 4); level_sync(l, crse_iteration); }
    }
    if (l > 0) mf_saxpy(lev[l]->p_avg, 1.0 / (double)n_cycle[l], lev[l]->P[lev[l]->pnew], 0, 0, 1, 0);      // incrPAvg
}

// Amr::timeStep
// Amr::timeStep's regrid block (upstream AMReX_Amr.cpp): at the start of a step of level l every level i = l .. min(finest, max_level - 1)
// whose own step count since the last rebuild has reached regrid_int rebuilds the levels above it
void AmrNS::maybe_regrid(int l, double time)
{
    if (!(rg.regrid_int > 0 && rg.max_level > 0)) return;
    if ((int)level_count_v.size() < rg.max_level + 1) level_count_v.resize(rg.max_level + 1, 0);
    int lev_top = std::min((int)lev.size() - 1, rg.max_level - 1);
    for (int i = l; i <= lev_top; ++i) {
        const int old_finest = (int)lev.size() - 1;
        if (level_count_v[i] >= rg.regrid_int) {
            AmrTimer t(*this, 5);
            std::vector<std::vector<BoxD>> grids = make_new_grids(i);
            const bool changed = install_grids(grids, i, time);
            if (changed) regrid_log.push_back(RegridEvent{i, time, grids});
            if (changed && i == 0 && rg.compute_new_dt_on_regrid) compute_new_dt(true);
            for (int k = i; k <= rg.max_level; ++k) level_count_v[k] = 0;
        }
        if (old_finest > (int)lev.size() - 1) lev_top = std::min((int)lev.size() - 1, rg.max_level - 1);
    }
}

void AmrNS::time_step(int l, double time, int iteration, int niter)
{
    maybe_regrid(l, time);
    NavierStokes& s = *lev[l];
    s.time = time;
    double dt_new;
    { AmrTimer t(*this, 8 + std::min(l, 7)); dt_new = s.advance(dt_level[l], iteration, niter); }
    dt_min[l] = iteration == 1 ? dt_new : std::min(dt_min[l], dt_new);
    s.time = time + dt_level[l];
    s.nstep += 1;
    if ((int)level_count_v.size() > l) level_count_v[l] += 1;
    if (l < (int)lev.size() - 1) {
        const int nc = n_cycle[l + 1];
        for (int i = 1; i <= nc; ++i) time_step(l + 1, time + (i - 1) * dt_level[l + 1], i, nc);
    }
    post_timestep(l, iteration);
}

// NavierStokes::post_init (NavierStokes.cpp:1254-1299) for the hierarchy; S_new of every level holds the initial data
void AmrNS::post_init(double stop_time_)
{
    auto& ctx = Context::get();
    const int nl = (int)lev.size(), fin = nl - 1;
    stop_time = stop_time_;
    for (auto& s : lev) {
        for (int q = 0; q < 2; ++q) { s->P[q].setVal(0.0); s->Gp[q].setVal(0.0); }
        s->time = 0.0; s->nstep = 0;
        s->set_time_level(0.0, 0.0, 0.0);
    }
    std::vector<MultiFab*> vel(nl), phi(nl);
    std::vector<const MultiFab*> sigp(nl);
    std::vector<MultiFab> sig(nl), vv(nl);
    std::vector<int> vcomp(nl, 0);
    const bool have_divu = lev[0]->have_divu;
    std::vector<MultiFab> rc(nl);
    std::vector<const MultiFab*> rcp(nl, nullptr);
    if (have_divu)                                                   // NavierStokes::initData (NavierStokes.cpp:457-479): rho at both times, divu, dsdt = 0
        for (auto& s : lev) {
            s->make_rho_curr_time();
            MultiFab::Copy(s->rho_ptime, s->rho_ctime, 0, 0, 1, 1);
            s->calc_divu(true);
            s->S[s->inew].setVal(0.0, s->Dsdt, 1, 0);
        }
    // ---- post_init_state (NavierStokesBase.cpp:2369-2439)
    if (p.init_vel_iter <= 0) { for (auto& s : lev) { s->P[1 - s->pnew].setVal(0.0); s->Gp[1 - s->pnew].setVal(0.0); } }
    else
    for (int iter = 0; iter < p.init_vel_iter; ++iter) {             // Projection::initialVelocityProject
        std::vector<ProjLevel> PL(nl);
        std::vector<const MultiFab*> dvp(nl, nullptr);
        std::vector<int> dvc(nl, 0);
        for (int l = 0; l < nl; ++l) {
            NavierStokes& s = *lev[l];
            PL[l] = proj_level(l);
            vel[l] = &s.S[s.inew]; vcomp[l] = Xvel; phi[l] = &s.P[1 - s.pnew];
            if (have_divu) { dvp[l] = &s.S[s.inew]; dvc[l] = s.Divu; }      // rhcc = -getDivCond(cur_divu_time), Projection.cpp:732-743, 783-788
        }
        lev[0]->st_nodal = initial_velocity_project(PL, vel.data(), vcomp.data(), phi.data(), nullptr /* rho_wgt_vel_proj = 0 */, nullptr,
                                                    have_divu ? dvp.data() : nullptr, dvc.data(), p.proj_tol, p.proj_abs_tol, o);
        for (auto& s : lev) for (int q = 0; q < 2; ++q) { s->P[q].setVal(0.0); s->Gp[q].setVal(0.0); }
    }
    for (auto& s : lev) s->initial_step = true;
    for (int l = fin - 1; l >= 0; --l) avg_down(l);
    if (std::abs(p.gravity) > 0.0) {                                  // Projection::initialPressureProject (Projection.cpp:841-960)
        for (int l = 0; l < nl; ++l) {
            NavierStokes& s = *lev[l];
            sig[l].define(s.layout, cell_type(), 1, 0);
            const FabD *st = sig[l].d_tab, *nt = s.S[s.inew].d_tab;
            for_each(*s.layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) { st[f](i, j, k) = 1.0 / nt[f](i, j, k, Density); });
            vv[l].define(s.layout, cell_type(), 3, 1);
            vv[l].setVal(0.0);
            vv[l].setVal(p.gravity, 2, 1, 1);
            vel[l] = &vv[l]; vcomp[l] = 0; phi[l] = &s.P[s.pnew]; sigp[l] = &sig[l];
        }
        // set_outflow_bcs(INITIAL_PRESS, c_lev = 0 .. f_lev; Projection.cpp:893-905, 1776-1803): the finest level that covers the whole strip of
        // an outflow face computes the hydrostatic data, putDown (:1656-1712) injects them into the coarser levels
        {
            bool done = false;
            for (int l = nl - 1; l >= 0; --l) {
                NavierStokes& s = *lev[l];
                if (!done) { done = s.set_outflow_bcs(*phi[l], s.S[s.inew], Density); continue; }
                MultiFab tmp(s.layout, node_type(), 1, 0);
                MultiFab::Copy(tmp, *phi[l], 0, 0, 1, 0);
                average_down(*phi[l + 1], tmp, 0, 1, lev[l + 1]->ratio);
                const FabD *pt = phi[l]->d_tab, *tt = tmp.d_tab;
                for (int D = 0; D < 2; ++D) for (int side = 0; side < 2; ++side) {
                    if (s.g.periodic[D] || (side == 0 ? s.p.phys_lo[D] : s.p.phys_hi[D]) != phys_outflow) continue;
                    const int face = side == 0 ? s.g.domain.lo[D] : s.g.domain.hi[D] + 1;
                    for_each(*s.layout, node_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) { if ((D == 0 ? i : j) == face) pt[f](i, j, k) = tt[f](i, j, k); });
                }
            }
        }
        lev[0]->st_nodal = composite_project(0, nl, vel.data(), vcomp.data(), phi.data(), sigp.data(), nullptr, p.proj_tol, p.proj_abs_tol, false, 0.0);
        for (auto& s : lev) { MultiFab::Copy(s->P[1 - s->pnew], s->P[s->pnew], 0, 0, 1, 1); MultiFab::Copy(s->Gp[1 - s->pnew], s->Gp[s->pnew], 0, 0, 3, 1); }
    }
    // ---- post_init_estDT (NavierStokesBase.cpp:2307-2362)
    std::vector<double> dt_save(nl);
    std::vector<int> nc_save(nl);
    double dt_init = 1.0e+100;
    for (int k = 0; k < nl; ++k) {
        nc_save[k] = n_cycle[k];
        dt_save[k] = p.init_shrink * lev[k]->estTimeStep();           // initialTimeStep
        int n_factor = 1;
        for (int m = fin; m > k; --m) n_factor *= n_cycle[m];
        dt_init = std::min(dt_init, dt_save[k] / (double)n_factor);
    }
    double dt0 = dt_save[0];
    { int n_factor = 1; for (int k = 0; k < nl; ++k) { n_factor *= nc_save[k]; dt0 = std::min(dt0, n_factor * dt_save[k]); } }
    if (stop_time >= 0.0) { const double eps = 0.0001 * dt0; if (0.0 + dt0 > stop_time - eps) dt0 = stop_time - 0.0; }
    { int n_factor = 1; for (int k = 0; k < nl; ++k) { n_factor *= nc_save[k]; dt_save[k] = dt0 / (double)n_factor; } }
    for (int k = 0; k < nl; ++k) { dt_level[k] = dt_init; n_cycle[k] = 1; lev[k]->set_time_level(0.0, dt_init, dt_init); }
    // ---- post_init_press (NavierStokes.cpp:1306-1432)
    if (p.init_iter > 0) {
        for (auto& s : lev) s->initial_iter = true;
        for (int iter = 0; iter < p.init_iter; ++iter) {
            for (int k = 0; k < nl; ++k) lev[k]->advance(dt_init, 1, 1);
            // Projection::initialSyncProject (Projection.cpp:970-1185)
            {
                std::vector<ProjLevel> PL(nl);
                std::vector<const MultiFab*> vold(nl), rh(nl), dn(nl, nullptr), dol(nl, nullptr);
                std::vector<MultiFab*> pn(nl);
                std::vector<int> dvc(nl, 0);
                for (int l = 0; l < nl; ++l) {
                    NavierStokes& s = *lev[l];
                    PL[l] = proj_level(l);
                    vel[l] = &s.S[s.inew]; vcomp[l] = Xvel; vold[l] = &s.S[1 - s.inew]; phi[l] = &s.P[1 - s.pnew]; pn[l] = &s.P[s.pnew]; rh[l] = &s.rho_half;
                    if (have_divu) { dn[l] = &s.S[s.inew]; dol[l] = &s.S[1 - s.inew]; dvc[l] = s.Divu; }
                }
                lev[0]->st_nodal = initial_sync_project(PL, vel.data(), vcomp.data(), vold.data(), phi.data(), pn.data(), rh.data(), have_divu ? dn.data() : nullptr,
                                                        have_divu ? dol.data() : nullptr, dvc.data(), dt_init, p.proj_tol, p.proj_abs_tol, o);
            }
            for (int k = fin - 1; k >= 0; --k) avg_down(k);
            for (auto& s : lev) {                                     // resetState(strt_time, dt_init, dt_init)
                s->inew = 1 - s->inew;
                if (s->have_divu) MultiFab::Copy(s->S[s->inew], s->S[1 - s->inew], s->Dsdt, s->Dsdt, 1, 0);   // Dsdt_Type is not reset (NavierStokesBase.cpp:2669-2676)
                MultiFab::Copy(s->P[1 - s->pnew], s->P[s->pnew], 0, 0, 1, 1);
                MultiFab::Copy(s->Gp[1 - s->pnew], s->Gp[s->pnew], 0, 0, 3, 1);
                s->set_time_level(0.0, dt_init, dt_init);
                s->initial_iter = false;
            }
        }
    }
    for (int k = 0; k < nl; ++k) {
        NavierStokes& s = *lev[k];
        s.initial_step = false; s.initial_iter = false;
        s.set_time_level(0.0, dt_save[k], dt_save[k]);
        dt_level[k] = dt_save[k]; n_cycle[k] = nc_save[k]; dt_min[k] = 1.e200;
        s.dt = dt_save[k];
    }
    level_steps = 0;
}

// NavierStokesBase::computeNewDt (NavierStokesBase.cpp:945-1035); post_regrid: limited by the pre-regrid dt instead of change_max x dt
void AmrNS::compute_new_dt(bool post_regrid)
{
    const int nl = (int)lev.size();
    const double cur_time = lev[0]->time;
    for (int i = 0; i < nl; ++i) dt_min[i] = std::min(dt_min[i], lev[i]->estTimeStep());
    if (p.fixed_dt <= 0.0) for (int i = 0; i < nl; ++i) dt_min[i] = std::min(dt_min[i], post_regrid ? dt_level[i] : p.change_max * dt_level[i]);
    double dt_0 = 1.0e+100;
    int n_factor = 1;
    for (int i = 0; i < nl; ++i) { n_factor *= n_cycle[i]; dt_0 = std::min(dt_0, n_factor * dt_min[i]); }
    const double eps = 0.0001 * dt_0;
    if (stop_time >= 0.0 && cur_time + dt_0 > stop_time - eps) dt_0 = stop_time - cur_time;
    n_factor = 1;
    for (int i = 0; i < nl; ++i) { n_factor *= n_cycle[i]; dt_level[i] = dt_0 / (double)n_factor; }
}

void AmrNS::get_restart_state(double* dt_lev, double* dt_mn, int* ncyc, int counters[2], double* stop) const
{
    for (size_t l = 0; l < lev.size(); ++l) { dt_lev[l] = dt_level[l]; dt_mn[l] = dt_min[l]; ncyc[l] = n_cycle[l]; }
    counters[0] = level_steps; counters[1] = level_count;
    *stop = stop_time;
}
// Amr::level_count of every level (steps of level i since the last regrid that started at or below it), sized max_level + 1 in a checkpoint
void AmrNS::get_level_counts(int* counts, int n) const
{
    for (int i = 0; i < n; ++i) counts[i] = i < (int)level_count_v.size() ? level_count_v[i] : 0;
    if (n > 0 && level_count_v.empty()) counts[0] = level_count;
}
void AmrNS::set_level_counts(const int* counts, int n)
{
    level_count_v.assign(counts, counts + n);
    if (n > 0) level_count = counts[0];
}
void AmrNS::set_restart_state(const double* dt_lev, const double* dt_mn, const int* ncyc, const int counters[2], double stop)
{
    for (size_t l = 0; l < lev.size(); ++l) { dt_level[l] = dt_lev[l]; dt_min[l] = dt_mn[l]; n_cycle[l] = ncyc[l]; lev[l]->dt = dt_lev[l]; }
    level_steps = counters[0]; level_count = counters[1];
    stop_time = stop;
}

// Amr::coarseTimeStep: computeNewDt + timeStep(0); Amr::timeStep regrids from level 0 at the start of the step once regrid_int coarse
// steps have been taken since the last regrid; only with amr.compute_new_dt_on_regrid = 1 (default 0) it then recomputes the time steps
// with post_regrid_flag = 1 -- otherwise levels that existed keep their dt and new levels start with dt_level[l-1] / n_cycle[l]
// (install_grids)
double AmrNS::coarse_step()
{
    if (level_steps > 0) compute_new_dt(false);
    regrid_log.clear();
    if ((int)level_count_v.size() < rg.max_level + 1) level_count_v.resize(rg.max_level + 1, 0);
    if (!level_count_v.empty()) level_count_v[0] = level_count;      // level_count: the value checkpoints carry (Amr::level_count[0])
    time_step(0, lev[0]->time, 1, 1);
    level_steps += 1;
    level_count = level_count_v.empty() ? level_count + 1 : level_count_v[0];
    for (size_t i = 0; i < lev.size(); ++i) lev[i]->dt = dt_level[i];
    return dt_level[0];
}

}  // namespace iamrx
