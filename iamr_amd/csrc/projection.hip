// iamr_amd/csrc/projection.hip -- nodal (approximate) projection and tensor-diffusion operator wrappers.
//
// nodal_projection mirrors Projection::doMLMGNodalProjection (reference Source/Projection.cpp:2385-2567)
// for one level: Hydro::NodalProjector computes rhs = div(vel) at nodes (velocity ghost cells filled
// first), solves div(sigma grad phi) = rhs with MLNodeLaplacian/MLMG, updates vel -= sigma grad phi, and
// Gradp is REPLACED (increment_gp = false: level / initial velocity projections) or INCREMENTED
// (true: sync projections) with grad phi (:2549-2563), followed by a ghost fill of Gradp (:2564-2565).
//
// tensor_apply / tensor_solve mirror the MLTensorOp set-up of Diffusion::getTensorViscTerms
// (Source/Diffusion.cpp:1655-1777) and Diffusion::diffuse_tensor_velocity (:650-957).
#include "operators.h"
#include "launch.h"
#include <cstdlib>

namespace iamrx {

// MLNodeLaplacian::compRHS, the cell-centred source (mlndlap_rhcc, then mlndlap_impose_neumann_bc on the sum with div(vel)):
// rhs(node) += 1/8 of the sum over the 8 cells around the node, doubled per Neumann / inflow wall direction the node lies on.
// rc: 1 ghost cell, zero outside the domain and on the cells that do not count (make_rhcc)
void nodal_rhcc_add(const Geometry& g, MultiFab& rhs, const MultiFab& rc, const DomainBC& bc)
{
    IAMRX_ASSERT(rc.ngrow >= 1);
    const FabD *rt = rhs.d_tab, *ct = rc.d_tab;
    const BoxD dom = g.domain;
    int nlo[3], nhi[3];
    for (int d = 0; d < 3; ++d) {
        nlo[d] = !g.periodic[d] && (bc.lo[d] == lo_neumann || bc.lo[d] == lo_inflow);
        nhi[d] = !g.periodic[d] && (bc.hi[d] == lo_neumann || bc.hi[d] == lo_inflow);
    }
    const int a0 = nlo[0], a1 = nlo[1], a2 = nlo[2], b0 = nhi[0], b1 = nhi[1], b2 = nhi[2];
    for_each(*rhs.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD c = ct[f];
        double s = 0.0;
        for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) s += c(i - 1 + cx, j - 1 + cy, k - 1 + cz);
        double r = 0.125 * s;
        if ((a0 && i == dom.lo[0]) || (b0 && i == dom.hi[0] + 1)) r *= 2.0;
        if ((a1 && j == dom.lo[1]) || (b1 && j == dom.hi[1] + 1)) r *= 2.0;
        if ((a2 && k == dom.lo[2]) || (b2 && k == dom.hi[2] + 1)) r *= 2.0;
        rt[f](i, j, k) += r;
    });
}

// scale * src(comp) on the valid cells (times the 0/1 complement of `drop` if given: cells where drop != 0 do not count), 1 ghost cell:
// neighbours and periodic images, zero elsewhere
MultiFab make_rhcc(const Geometry& g, const MultiFab& src, int comp, double scale, const MultiFab* drop)
{
    MultiFab rc(src.layout, cell_type(), 1, 1);
    rc.setVal(0.0);
    MultiFab::Copy(rc, src, comp, 0, 1, 0);
    if (scale != 1.0) mf_mult(rc, scale, 0, 1, 0);
    if (drop) mask_mult(rc, 0, 1, *drop, true, 0);
    rc.FillBoundary(g);
    return rc;
}

MGStats nodal_projection(const Geometry& g, MultiFab& vel, int vcomp, MultiFab& phi, const MultiFab& sig, int sig_comp,
                         const DomainBC& bc, double rel_tol, double abs_tol, const MGOpts& opts, MultiFab* gp, bool increment_gp, const MultiFab* rhcc)
{
    LayoutP layout = phi.layout;
    // caller-owned arrays on a level chopped at amr.max_grid_size: project on the merged boxes (see mlmg_mac_solve); a level with a
    // Dirichlet mask from partial coverage never qualifies (merged_solve_layout: the boxes cover the domain)
    if (LayoutP ml = merged_solve_layout(g, layout)) {
        MultiFab vel_m(ml, cell_type(), 3, vel.ngrow), phi_m(ml, node_type(), 1, phi.ngrow), sig_m(ml, cell_type(), 1, sig.ngrow), gp_m, rh_m;
        relayout_copy(vel_m, vel, 3, vcomp, 0);
        relayout_copy(phi_m, phi, 1);
        relayout_copy(sig_m, sig, 1, sig_comp, 0);
        if (gp) { gp_m.define(ml, cell_type(), 3, gp->ngrow); if (increment_gp) relayout_copy(gp_m, *gp, 3); }
        if (rhcc) { rh_m.define(ml, cell_type(), 1, rhcc->ngrow); relayout_copy(rh_m, *rhcc, 1); }
        MGStats st = nodal_projection(g, vel_m, 0, phi_m, sig_m, 0, bc, rel_tol, abs_tol, opts, gp ? &gp_m : nullptr, increment_gp, rhcc ? &rh_m : nullptr);
        relayout_copy(vel, vel_m, 3, 0, vcomp);
        relayout_copy(phi, phi_m, 1);
        if (gp) relayout_copy(*gp, gp_m, 3);
        return st;
    }
    // set_boundary_velocity + periodic fill of the velocity ghost cells
    vel.FillBoundary(g, vcomp, 3);
    NodalMG mg(g, layout, bc, opts);
    mg.setSigma(sig, sig_comp);
    MultiFab rhs(layout, node_type(), 1, 0);
    nodal_divu(g, rhs, vel, vcomp, &bc);
    if (rhcc) nodal_rhcc_add(g, rhs, *rhcc, bc);
    MGStats st = mg.solve(phi, rhs, rel_tol, abs_tol);
    nodal_mknewu(g, &vel, vcomp, phi, &mg.sigma(0), gp, increment_gp);
    if (gp) gp->FillBoundary(g);
    return st;
}

MGStats level_project_single(const Geometry& g, double dt, MultiFab& U_new, int vcomp, MultiFab& P_new, const MultiFab& Gp_old, MultiFab& Gp_new,
                             const MultiFab& rho_half, const DomainBC& bc, double proj_tol, double proj_abs_tol, const MGOpts& opts)
{
    LayoutP layout = P_new.layout;
    IAMRX_ASSERT(layout->total_cells() == g.domain.npts());
    P_new.setVal(0.0, 0, 1, 0);                                  // Projection.cpp:236-256
    mf_mult(U_new, 1.0 / dt, vcomp, 3, 1);                       // :273
    MultiFab sig(layout, cell_type(), 1, 1);
    {
        const FabD *nt = U_new.d_tab, *gt = Gp_old.d_tab, *ht = rho_half.d_tab, *st = sig.d_tab;
        for_each(*layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
            const double rh = ht[f](i, j, k, 0);
            for (int n = 0; n < 3; ++n) nt[f](i, j, k, vcomp + n) += gt[f](i, j, k, n) / rh;       // :296-300
            st[f](i, j, k, 0) = 1.0 / rh;                                                            // scaleVar
        });
    }
    MGStats st = nodal_projection(g, U_new, vcomp, P_new, sig, 0, bc, proj_tol, proj_abs_tol, opts, &Gp_new, false);
    mf_mult(U_new, dt, vcomp, 3, 1);                             // :438
    return st;
}

static void setup_tensor(CellMG& mg, MultiFab tb[3], const MultiFab* bp[3], LayoutP layout, double a_scalar, double b_scalar,
                         const MultiFab* acoef, const MultiFab* const eta[3])
{
    const bool eta_form = tune("TENSOR_ETA", 1) != 0;
    for (int d = 0; d < 3; ++d) {
        if (eta_form) { bp[d] = eta[d]; continue; }
        tb[d].define(layout, face_type(d), 3, 0);
        tensor_bcoef(tb[d], *eta[d], d);
        bp[d] = &tb[d];
    }
    mg.setTensorEta(eta_form);
    mg.setScalars(a_scalar, b_scalar);
    if (acoef) mg.setACoeffs(acoef);
    mg.setBCoeffs(bp);
    mg.setTensor(true);
}

void tensor_apply(const Geometry& g, MultiFab& out, MultiFab& vel, double a_scalar, double b_scalar, const MultiFab* acoef,
                  const MultiFab* const eta[3], const DomainBC* bcs, int nbc, const TensorCF* cf, const TensorFlux* fx)
{
    MGOpts o;
    o.max_coarsening_level = 0;      // info.setMaxCoarseningLevel(0) (Diffusion.cpp:708)
    o.maxorder = bcs[0].maxorder;
    CellMG mg(g, vel.layout, 3, bcs[0], o);
    if (nbc > 1) mg.setDomainBCs(bcs, nbc);
    MultiFab tb[3];
    const MultiFab* bp[3];
    setup_tensor(mg, tb, bp, vel.layout, a_scalar, b_scalar, acoef, eta);
    if (cf) mg.setCoarseFineBC(cf->crse, *cf->cgeom, cf->ratio);      // tensorop.setCoarseFineBC (Diffusion.cpp:1725-1736)
    mg.prepare();
    mg.apply(out, vel);
    if (fx) tensor_extensive_flux(g, vel, eta, fx->flux, fx->fac, fx->add);      // computeExtensiveFluxes after mlmg.apply (Diffusion.cpp:790-796)
}

MGStats tensor_solve(const Geometry& g, MultiFab& soln, const MultiFab& rhs, double a_scalar, double b_scalar, const MultiFab* acoef,
                     const MultiFab* const eta[3], const DomainBC* bcs, int nbc, double tol_rel, double tol_abs, const MGOpts& opts, const TensorCF* cf,
                     const TensorFlux* fx)
{
    MGOpts o = opts;
    o.maxorder = bcs[0].maxorder;
    CellMG mg(g, soln.layout, 3, bcs[0], o);
    if (nbc > 1) mg.setDomainBCs(bcs, nbc);
    MultiFab tb[3];
    const MultiFab* bp[3];
    setup_tensor(mg, tb, bp, soln.layout, a_scalar, b_scalar, acoef, eta);
    if (cf) mg.setCoarseFineBC(cf->crse, *cf->cgeom, cf->ratio);      // Diffusion.cpp:876-887 (crse == null: :1096-1099)
    mg.prepare();
    MGStats st = mg.solve(soln, rhs, tol_rel, tol_abs);        // ends with the ghost cells of soln filled by the operator (setFinalFillBC)
    if (fx) tensor_extensive_flux(g, soln, eta, fx->flux, fx->fac, fx->add);
    return st;
}

}  // namespace iamrx
