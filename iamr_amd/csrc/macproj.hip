// iamr_amd/csrc/macproj.hip -- MAC (face velocity) projection.
// Mirrors MacProj::mlmg_mac_solve (reference Source/MacProj.cpp:1084-1184) and the Hydro::MacProjector
// it drives: beta_face = average_cellcenter_to_face(rho) -> invert(1/rhs_scale) -> FillBoundary
// (:1115-1127); rhs = S - div(u_mac); solve -div(beta grad phi) = rhs to (mac_tol, mac_abs_tol) with
// max_order 4 (:1172); u_mac += -beta grad phi (getFluxes, :1181-1183).
#include "operators.h"
#include "launch.h"

namespace iamrx {

LayoutP merged_solve_layout(const Geometry& g, const LayoutP& l)
{
    if (!l || l->replicated || l->total_cells() != g.domain.npts()) return nullptr;
    LayoutP ml = coalesce_layout(l);
    return (ml->id != l->id && ml->boxes.size() < l->boxes.size()) ? ml : nullptr;
}

MGStats mlmg_mac_solve(const Geometry& g, MultiFab* const umac[3], const MultiFab& rho, int rho_comp, const MultiFab* S,
                       MultiFab& mac_phi, double rhs_scale, const DomainBC& bc, double mac_tol, double mac_abs_tol,
                       const MGOpts& opts, MultiFab* const fluxes[3], const MultiFab* cphi, const Geometry* cgeom, int ratio)
{
    LayoutP layout = mac_phi.layout;
    // caller-owned arrays on a level chopped at amr.max_grid_size (the operator boundary as IAMR drives it): solve on the merged boxes --
    // no ghost fills between colour passes, index wrap on periodic domains -- and hand the results back on the caller's
    if (!cgeom) {
        if (LayoutP ml = merged_solve_layout(g, layout)) {
            MultiFab um_m[3], fl_m[3], rho_m(ml, cell_type(), 1, std::min(rho.ngrow, 1)), phi_m(ml, cell_type(), 1, mac_phi.ngrow), S_m;
            MultiFab *ump[3], *flp[3];
            for (int d = 0; d < 3; ++d) {
                um_m[d].define(ml, face_type(d), 1, 0); relayout_copy(um_m[d], *umac[d], 1); ump[d] = &um_m[d];
                if (fluxes) { fl_m[d].define(ml, face_type(d), 1, 0); flp[d] = &fl_m[d]; }
            }
            relayout_copy(rho_m, rho, 1, rho_comp, 0);
            relayout_copy(phi_m, mac_phi, 1);
            if (S) { S_m.define(ml, cell_type(), 1, 0); relayout_copy(S_m, *S, 1); }
            MGStats st = mlmg_mac_solve(g, ump, rho_m, 0, S ? &S_m : nullptr, phi_m, rhs_scale, bc, mac_tol, mac_abs_tol, opts, fluxes ? flp : nullptr,
                                        nullptr, nullptr, ratio);
            relayout_copy(mac_phi, phi_m, 1);
            for (int d = 0; d < 3; ++d) { relayout_copy(*umac[d], um_m[d], 1); if (fluxes) relayout_copy(*fluxes[d], fl_m[d], 1); }
            return st;
        }
    }
    MultiFab bcoef[3];
    MultiFab* bp[3];
    const MultiFab* bcp[3];
    for (int d = 0; d < 3; ++d) { bcoef[d].define(layout, face_type(d), 1, 0); bp[d] = &bcoef[d]; bcp[d] = &bcoef[d]; }
    mac_bcoef(bp, rho, rho_comp, 1.0 / rhs_scale);

    MultiFab rhs(layout, cell_type(), 1, 0);
    const MultiFab* um[3] = {umac[0], umac[1], umac[2]};
    mac_rhs(g, rhs, um, S);

    CellMG mg(g, layout, 1, bc, opts);
    mg.setScalars(0.0, 1.0);
    mg.setBCoeffs(bcp);
    if (rho.ngrow >= 1) mg.setBCoeffsFromCell(&rho, rho_comp, 1.0 / rhs_scale);
    // level > 0: Dirichlet data on the coarse/fine faces from the coarse level's MAC phi (MacProj.cpp:1166-1170)
    if (cgeom) mg.setCoarseFineBC(cphi, *cgeom, ratio);
    mg.prepare();
    MGStats st = mg.solve(mac_phi, rhs, mac_tol, mac_abs_tol);
    // u_mac += (-beta grad phi); optionally hand the fluxes back (MacProj.cpp:1181-1183)
    mg.fluxes(mac_phi, fluxes, umac);
    return st;
}

// MacProj::mac_sync_solve (Source/MacProj.cpp:359-470): the coarse-level correction of the MAC velocity for the mismatch between the
// coarse and the (time-averaged) fine face velocities on the coarse/fine interface.  mr: the level's mac register after
// CrseInit(u_mac * area, -1) and the FineAdd(u_mac * area, 1/ncycle) of the fine sub-steps.
//   Rhs = Reflux(mr, scale -1) on the coarse cells next to the fine grids, 0 under them;  Rhs.negate();
//   solve -div(b grad phi) = Rhs with b = (dt/2)/rho_half on faces, no velocity;  Ucorr = -(-b grad phi)
MGStats mac_sync_solve(const Geometry& g, FluxRegister& mr, const MultiFab& rho_half, double dt, LayoutP fine_layout, int ratio,
                       MultiFab* const Ucorr[3], MultiFab& mac_sync_phi, const DomainBC& bc, double tol, double abs_tol, const MGOpts& opts,
                       const Geometry* cgeom, int cratio)
{
    LayoutP layout = mac_sync_phi.layout;
    MultiFab Rhs(layout, cell_type(), 1, 0);
    Rhs.setVal(0.0);
    mr.Reflux(Rhs, g.dx[0] * g.dx[1] * g.dx[2], -1.0, 0, 0, 1);
    {
        MultiFab fz(fine_layout, cell_type(), 1, 0);      // zero under the fine grids (MacProj.cpp:395-415)
        fz.setVal(0.0);
        average_down(fz, Rhs, 0, 1, ratio);
    }
    mac_sync_phi.setVal(0.0);
    mf_mult(Rhs, -1.0, 0, 1, 0);
    MultiFab um[3];
    MultiFab* ump[3];
    for (int d = 0; d < 3; ++d) { um[d].define(layout, face_type(d), 1, 0); um[d].setVal(0.0); ump[d] = &um[d]; }
    // on a refined level the coarse/fine faces carry homogeneous Dirichlet data (cphi = null, MacProj.cpp:454-456)
    MGStats st = mlmg_mac_solve(g, ump, rho_half, 0, &Rhs, mac_sync_phi, 2.0 / dt, bc, tol, abs_tol, opts, Ucorr, nullptr, cgeom, cratio);
    for (int d = 0; d < 3; ++d) mf_mult(*Ucorr[d], -1.0, 0, 1, 0);
    return st;
}

}  // namespace iamrx
