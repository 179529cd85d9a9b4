// iamr_amd/csrc/diffusion.hip -- the Diffusion operator entries on CALLER-OWNED data (reference Source/Diffusion.H:53-225): what
// NavierStokes::scalar_diffusion_update / velocity_diffusion_update / mac_sync hand to the Diffusion class, as free functions of plain
// MultiFabs, so that a host code that keeps its own state (the operator-level boundary, C-ABI iamrx_diffuse_*) can call the
// implicit viscous / diffusive updates without an iamrx NavierStokes level.  iamrx::NavierStokes itself calls these.
//   diffuse_scalar            Diffusion::diffuse_scalar           Source/Diffusion.cpp:207-599
//   diffuse_tensor_velocity   Diffusion::diffuse_tensor_velocity  Source/Diffusion.cpp:617-957
//   diffuse_tensor_Vsync      Diffusion::diffuse_tensor_Vsync     Source/Diffusion.cpp:1010-1178
//   diffuse_Ssync             Diffusion::diffuse_Ssync            Source/Diffusion.cpp:1181-1352
#include "operators.h"
#include "launch.h"

namespace iamrx {

void scale_by(MultiFab& y, const MultiFab& x, int xcomp, int ng, bool divide);

namespace {
// y(ycomp) *= x(xcomp) or /= on the valid cells + ng
void scale_comp(MultiFab& y, int ycomp, const MultiFab& x, int xcomp, int ng, bool divide)
{
    const FabD *yt = y.d_tab, *xt = x.d_tab;
    for_each(*y.layout, cell_type(), ng, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const double x = xt[f](i, j, k, xcomp);
        if (!divide) yt[f](i, j, k, ycomp) *= x;
        else if (x != 0.0 || yt[f](i, j, k, ycomp) != 0.0) yt[f](i, j, k, ycomp) /= x;     // 0 / 0 (an unfilled ghost cell of a sync increment) stays 0
    });
}
// the coarse level's component sigma (over its density for rho_flag 2) on the coarse layout: Solnc of Diffusion.cpp:376-396, 506-518
void crse_soln(MultiFab& out, const MultiFab& Sc, int sigma, int rho_comp, bool over_rho)
{
    out.define(Sc.layout, cell_type(), 1, 0);
    MultiFab::Copy(out, Sc, sigma, 0, 1, 0);
    if (over_rho) scale_comp(out, 0, Sc, rho_comp, 0, true);
}
}  // namespace

MGStats diffuse_scalar(const Geometry& g, const MultiFab* S_old, const MultiFab* Rho_old, MultiFab& S_new, const MultiFab* Rho_new, int sigma, int rho_comp, double dt, double theta,
                       const MultiFab& rho_half, int rho_flag, MultiFab* const fluxn[3], MultiFab* const fluxnp1[3],
                       const MultiFab* delta_rhs, int rhs_comp, const MultiFab* const betan[3], const MultiFab* const betanp1[3],
                       const DomainBC& bc, const DiffusionCrse* crse, bool add_old_time_divFlux, double visc_tol, const MGOpts& o)
{
    LayoutP layout = S_new.layout;
    IAMRX_ASSERT(S_new.ngrow >= 1 && (rho_flag >= 0 && rho_flag <= 2));
    const bool cons = rho_flag == 2;
    if (!Rho_old) Rho_old = S_old;
    if (!Rho_new) Rho_new = &S_new;
    const bool old_part = add_old_time_divFlux && theta != 1.0;
    IAMRX_ASSERT(!old_part || (S_old && S_old->ngrow >= 1 && betan));
    MultiFab Rhs(layout, cell_type(), 1, 0);
    const bool want_flux = fluxn != nullptr && fluxnp1 != nullptr && fluxn[0] != nullptr;
    MultiFab cdata;
    if (old_part) {
        // Soln = S_old (/ rho_old) with its ghost cells as level BC (Diffusion.cpp:355-413); a = 0, b = -(1 - theta) dt
        MultiFab Soln0(layout, cell_type(), 1, 1);
        MultiFab::Copy(Soln0, *S_old, sigma, 0, 1, 1);
        if (cons) scale_comp(Soln0, 0, *Rho_old, rho_comp, 1, true);
        MGOpts mo;
        mo.max_coarsening_level = 0;                             // infon.setMaxCoarseningLevel(0) (Diffusion.cpp:318)
        mo.maxorder = 2;
        CellMG opn(g, layout, 1, bc, mo);
        opn.setScalars(0.0, -(1.0 - theta) * dt);
        opn.setBCoeffs(betan);
        if (crse) { crse_soln(cdata, *crse->crse_old, sigma, rho_comp, cons); opn.setCoarseFineBC(&cdata, *crse->cgeom, crse->ratio); }
        opn.prepare();
        opn.apply(Rhs, Soln0);
        if (want_flux) {                                         // fluxn = (1 - theta) * area * (-beta grad s_old) (Diffusion.cpp:437-438)
            opn.fluxes(Soln0, fluxn, nullptr);
            for (int d = 0; d < 3; ++d) mf_mult(*fluxn[d], (1.0 - theta) * g.dx[(d + 1) % 3] * g.dx[(d + 2) % 3] / (-(1.0 - theta) * dt), 0, 1, 0);
        }
    } else {
        Rhs.setVal(0.0);
        if (want_flux) for (int d = 0; d < 3; ++d) fluxn[d]->setVal(0.0);
    }
    {   // body sources, then rhs += S_new (x rho_half for rho_flag 1): Diffusion.cpp:455-487
        const FabD *rt = Rhs.d_tab, *st = S_new.d_tab, *ht = rho_half.d_tab, *dtab = delta_rhs ? delta_rhs->d_tab : nullptr;
        const int rf = rho_flag;
        for_each(*layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
            double r = rt[f](i, j, k, 0);
            if (dtab) { r += dtab[f](i, j, k, rhs_comp) * dt; if (rf == 1) r *= ht[f](i, j, k, 0); }
            double s = st[f](i, j, k, sigma);
            if (rf == 1) s *= ht[f](i, j, k, 0);
            rt[f](i, j, k, 0) = r + s;
        });
    }
    const double tol_abs = visc_tol * Rhs.norm0(0, 1, 0);        // get_scaled_abs_tol (Diffusion.cpp:193-204)
    MultiFab Soln(layout, cell_type(), 1, 1);
    MultiFab::Copy(Soln, S_new, sigma, 0, 1, 1);                 // initial guess + level BC: S_new with its ghost cells
    MultiFab acoef(layout, cell_type(), 1, 0);
    acoef.setVal(1.0);                                           // computeAlpha (Diffusion.cpp:1355-1398)
    if (cons) { scale_comp(Soln, 0, *Rho_new, rho_comp, 1, true); MultiFab::Copy(acoef, *Rho_new, rho_comp, 0, 1, 0); }   // :520-540
    else if (rho_flag == 1) MultiFab::Copy(acoef, rho_half, 0, 0, 1, 0);
    MGOpts so = o;
    so.maxorder = 2;                                             // Diffusion::max_order
    CellMG opnp1(g, layout, 1, bc, so);
    opnp1.setScalars(1.0, theta * dt);
    opnp1.setACoeffs(&acoef);
    opnp1.setBCoeffs(betanp1);
    if (crse) {                                                  // Diffusion.cpp:506-518; sync solves pass no coarse data: homogeneous
        if (crse->crse_new) { crse_soln(cdata, *crse->crse_new, sigma, rho_comp, cons); opnp1.setCoarseFineBC(&cdata, *crse->cgeom, crse->ratio); }
        else opnp1.setCoarseFineBC(nullptr, *crse->cgeom, crse->ratio);
    }
    opnp1.prepare();
    MGStats st = opnp1.solve(Soln, Rhs, visc_tol, tol_abs);
    if (want_flux) {                                             // fluxnp1 = theta * area * (-beta grad s_new) (Diffusion.cpp:569-570)
        opnp1.fluxes(Soln, fluxnp1, nullptr);
        for (int d = 0; d < 3; ++d) mf_mult(*fluxnp1[d], theta * g.dx[(d + 1) % 3] * g.dx[(d + 2) % 3] / (theta * dt), 0, 1, 0);
    }
    if (cons) scale_comp(Soln, 0, *Rho_new, rho_comp, 0, false); // Diffusion.cpp:583-590
    MultiFab::Copy(S_new, Soln, 0, sigma, 1, 0);
    return st;
}

MGStats diffuse_tensor_velocity(const Geometry& g, const MultiFab* U_old, MultiFab& U_new, int rho_comp, double dt, double theta,
                                const MultiFab& rho_half, int rho_flag, const MultiFab* visc_old_term, const MultiFab* const eta_n[3],
                                const MultiFab* const eta_np1[3], const DomainBC bc_visc[3], const DiffusionCrse* crse,
                                MultiFab* const tflux[3], double visc_tol, const MGOpts& o, const std::function<void(MultiFab&)>& fill_new)
{
    LayoutP layout = U_new.layout;
    // The solve runs IN PLACE on the velocity components of U_new and reads alpha from the array that holds it (views, below): U_new comes
    // with at least one ghost layer (level boundary data / initial guess); any wider ghost region is simply not touched (the kernels index
    // through the array descriptors; only the first layer is filled and read).  rho_half is read on the valid cells.
    IAMRX_ASSERT(U_new.ngrow >= 1 && (rho_flag == 1 || rho_flag == 3));
    MultiFab Rhs(layout, cell_type(), 3, 0);
    const bool want_flux = tflux != nullptr && tflux[0] != nullptr;
    TensorFlux fx{{want_flux ? tflux[0] : nullptr, want_flux ? tflux[1] : nullptr, want_flux ? tflux[2] : nullptr}, 0.0, false};
    if (want_flux) for (int d = 0; d < 3; ++d) tflux[d]->setVal(0.0);
    MultiFab cdata;
    TensorCF cf{&cdata, crse ? crse->cgeom : nullptr, crse ? crse->ratio : 2};
    // (1 - theta) dt div tau(U^n) from viscous terms the caller has evaluated already (no fluxes wanted): formed inside the pass below
    // (the expression of mf_lincomb(Rhs, (1 - theta) dt, visc, 0.0, visc), then Rhs += rho u*: the same doubles, one pass less)
    const bool from_visc = theta != 1.0 && !want_flux && visc_old_term;
    if (!from_visc && theta != 1.0) {
        IAMRX_ASSERT(U_old && U_old->ngrow >= 1);
        MultiFab Soln0(layout, cell_type(), 3, 1);
        MultiFab::Copy(Soln0, *U_old, Xvel, 0, 3, 1);
        if (crse) { cdata.define(crse->crse_old->layout, cell_type(), 3, 0); MultiFab::Copy(cdata, *crse->crse_old, Xvel, 0, 3, 0); }   // Diffusion.cpp:733-744
        fx.fac = 1.0 - theta; fx.add = false;                    // computeExtensiveFluxes(..., -b/dt), b = -(1 - theta) dt
        tensor_apply(g, Rhs, Soln0, 0.0, -(1.0 - theta) * dt, nullptr, eta_n, bc_visc, 3, crse ? &cf : nullptr, want_flux ? &fx : nullptr);
    } else if (!from_visc) Rhs.setVal(0.0);
    {
        const FabD *nt = U_new.d_tab, *ot = U_old ? U_old->d_tab : nullptr, *rt = Rhs.d_tab, *ht = rho_half.d_tab;
        const FabD* vt = from_visc ? visc_old_term->d_tab : nullptr;
        const double va = (1.0 - theta) * dt;
        const bool mom = rho_flag == 3;                          // rho_flag 3 (NavierStokes.cpp:1016): the OLD density (Diffusion.cpp:819)
        IAMRX_ASSERT(!mom || ot);
        for_each(*layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
            const double r = mom ? ot[f](i, j, k, rho_comp) : ht[f](i, j, k, 0);
            for (int n = 0; n < 3; ++n) {
                const double un = nt[f](i, j, k, n) * r;
                nt[f](i, j, k, n) = un;                          // Diffusion.cpp:825: the state holds rho u* from here on
                double rr;
                if (vt) { const double v = vt[f](i, j, k, n); rr = va * v + 0.0 * v; }
                else rr = rt[f](i, j, k, n);
                rt[f](i, j, k, n) = rr + un;
            }
        });
    }
    double avg = 0.0;                                            // get_scaled_abs_tol (Diffusion.cpp:193-204)
    double rn3[3];
    Rhs.norm0_comps(0, 3, 0, rn3);
    for (int n = 0; n < 3; ++n) avg += (1.0 / 3.0) * rn3[n];
    const double tol_abs = visc_tol * avg;
    // Diffusion.cpp:866: FillPatch(U_new) -- of the velocity components that hold rho u* by now: neighbours and periodic images carry rho u*,
    // the physical boundary functor and the coarse level their plain velocities (as written upstream).  The caller's fill does that.
    if (fill_new) fill_new(U_new);
    // Soln (initial guess + level BC = U_new with its ghost cells, Diffusion.cpp:866-870; copied back at :928) is the velocity components
    // of U_new themselves, alpha = rho_new (rho_flag 3, :893) or rho_half (computeAlpha, rho_flag 1) the arrays that hold them: no copies
    MultiFab Soln, acoef;
    Soln.view_of(U_new, Xvel, 3);
    if (rho_flag == 3) acoef.view_of(U_new, rho_comp, 1);
    else acoef.view_of(rho_half, 0, 1);
    MGOpts vo = o;
    vo.maxorder = 2;
    if (crse) { cdata.define(crse->crse_new->layout, cell_type(), 3, 0); MultiFab::Copy(cdata, *crse->crse_new, Xvel, 0, 3, 0); }      // Diffusion.cpp:876-887
    fx.fac = theta; fx.add = true;                               // computeExtensiveFluxes(..., b/dt) added to the old-time fluxes (:941-945)
    MGStats st = tensor_solve(g, Soln, Rhs, 1.0, theta * dt, &acoef, eta_np1, bc_visc, 3, visc_tol, tol_abs, vo, crse ? &cf : nullptr, want_flux ? &fx : nullptr);
    return st;
}


// Diffusion::diffuse_Vsync -> diffuse_tensor_Vsync (Diffusion.cpp:960-1178): (alpha - theta dt div tau) Vsync' = rho Vsync with homogeneous
// boundary and coarse/fine data; rho = rho_half (rho_flag 1) or the old density with alpha = the new one (rho_flag 3).  NOTE the face
// coefficients of this solve are set to 1.0 upstream (:1122-1135), not to the viscosity: `eta` is what the caller passes (iamrx's mac_sync
// passes ones, as written upstream).  Afterwards the ghost cells outside ext_dir faces are zero (:987-1008).  tflux: theta area (-tau).
MGStats diffuse_tensor_Vsync(const Geometry& g, MultiFab& Vsync, double dt, double theta, const MultiFab& rho_half, int rho_flag,
                             const MultiFab* Rho_old, const MultiFab* Rho_new, int rho_comp, const MultiFab* const eta[3],
                             const DomainBC bc_visc[3], const BCRec bc_vel[3], const Geometry* cgeom, int ratio, MultiFab* const tflux[3],
                             double visc_tol, const MGOpts& o)
{
    auto& ctx = Context::get();
    LayoutP layout = Vsync.layout;
    IAMRX_ASSERT(Vsync.ngrow >= 1 && (rho_flag == 1 || (rho_flag == 3 && Rho_old && Rho_new)));
    const bool rf3 = rho_flag == 3;
    MultiFab Rhs(layout, cell_type(), 3, 0), acoef(layout, cell_type(), 1, 0), Soln(layout, cell_type(), 3, 1);
    MultiFab::Copy(Rhs, Vsync, 0, 0, 3, 0);
    {
        const FabD *rt = Rhs.d_tab, *ht = rho_half.d_tab, *ot = rf3 ? Rho_old->d_tab : nullptr;
        for_each(*layout, cell_type(), 0, ctx.stream, [=] __device__(int i, int j, int k, int fb) {
            const double r = rf3 ? ot[fb](i, j, k, rho_comp) : ht[fb](i, j, k, 0);
            for (int n = 0; n < 3; ++n) rt[fb](i, j, k, n) *= r;
        });
    }
    if (rf3) MultiFab::Copy(acoef, *Rho_new, rho_comp, 0, 1, 0); else MultiFab::Copy(acoef, rho_half, 0, 0, 1, 0);
    Soln.setVal(0.0);
    MGOpts vo = o;
    vo.maxorder = 2;
    TensorCF cf{nullptr, cgeom, ratio};
    const bool want_flux = tflux != nullptr && tflux[0] != nullptr;
    TensorFlux fx{{want_flux ? tflux[0] : nullptr, want_flux ? tflux[1] : nullptr, want_flux ? tflux[2] : nullptr}, theta, false};
    MGStats st = tensor_solve(g, Soln, Rhs, 1.0, theta * dt, &acoef, eta, bc_visc, 3, visc_tol, -1.0, vo, cgeom ? &cf : nullptr, want_flux ? &fx : nullptr);
    MultiFab::Copy(Vsync, Soln, 0, 0, 3, 1);
    for (int n = 0; n < 3; ++n) for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
        if (g.periodic[d] || (side == 0 ? bc_vel[n].lo[d] : bc_vel[n].hi[d]) != bc_ext_dir) continue;
        const int face = side == 0 ? g.domain.lo[d] - 1 : g.domain.hi[d] + 1;
        const FabD* vt = Vsync.d_tab;
        const int dd = d, nn = n;
        for_each(*layout, cell_type(), 1, ctx.stream, [=] __device__(int i, int j, int k, int fb) {
            if ((dd == 0 ? i : (dd == 1 ? j : k)) == face) vt[fb](i, j, k, nn) = 0.0;
        });
    }
    return st;
}

// Diffusion::diffuse_Ssync as NavierStokes::mac_sync calls it (NavierStokes.cpp:1590-1640: diffuse_scalar with S_old = {}, S_new = 0,
// delta_rhs = Ssync, no old-time flux): component sn of Ssync (a rate) becomes the diffused sync increment,
//   (alpha - theta dt div beta grad) s = dt Ssync (x rho_half for rho_flag 1), Ssync = s (x rho_new for rho_flag 2);
// homogeneous boundary and coarse/fine data.  flux (may be null): theta area (-beta grad s).
MGStats diffuse_Ssync(const Geometry& g, MultiFab& Ssync, int sn, double dt, double theta, const MultiFab& rho_half, int rho_flag,
                      const MultiFab& Rho_new, int rho_comp, const MultiFab* const beta[3], const DomainBC& bc, const Geometry* cgeom, int ratio,
                      MultiFab* const flux[3], double visc_tol, const MGOpts& o)
{
    MultiFab dS(Ssync.layout, cell_type(), 1, 1);
    dS.setVal(0.0);
    MultiFab fn[3];
    MultiFab* fnp[3] = {&fn[0], &fn[1], &fn[2]};
    const bool want_flux = flux != nullptr && flux[0] != nullptr;
    if (want_flux) for (int d = 0; d < 3; ++d) fn[d].define(Ssync.layout, face_type(d), 1, 0);
    DiffusionCrse dc{nullptr, nullptr, cgeom, ratio};
    MGStats st = diffuse_scalar(g, nullptr, nullptr, dS, &Rho_new, 0, rho_comp, dt, theta, rho_half, rho_flag, want_flux ? fnp : nullptr, want_flux ? flux : nullptr,
                                &Ssync, sn, nullptr, beta, bc, cgeom ? &dc : nullptr, false, visc_tol, o);
    MultiFab::Copy(Ssync, dS, 0, sn, 1, 0);
    return st;
}

}  // namespace iamrx
