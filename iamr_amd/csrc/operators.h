// iamr_amd/csrc/operators.h -- host-side operator layer keeping IAMR's operator API surface
// (MacProj / Projection / Diffusion / NavierStokesBase::advance), SURVEY 8(b).
#pragma once
#include <string>
#include "mf.h"
#include "mlmg.h"
#include "kernels.h"
#include <memory>
#include <functional>

namespace iamrx {

// ---- MacProj (reference Source/MacProj.H:36-94) ------------------------------------------------------
// MacProj::mlmg_mac_solve (Source/MacProj.cpp:1084-1184)
MGStats mlmg_mac_solve(const Geometry& g, MultiFab* const umac[3], const MultiFab& rho, int rho_comp, const MultiFab* S,
                       MultiFab& mac_phi, double rhs_scale, const DomainBC& bc, double mac_tol, double mac_abs_tol,
                       const MGOpts& opts, MultiFab* const fluxes[3], const MultiFab* cphi = nullptr, const Geometry* cgeom = nullptr, int ratio = 2);

// ---- Projection (reference Source/Projection.H:53-134, 244-254) --------------------------------------
// Projection::doMLMGNodalProjection (Source/Projection.cpp:2385-2567), single level:
// rhs = div(vel) at nodes, solve div(sig grad phi) = rhs, vel -= sig grad phi, Gp = / += grad phi.
MGStats nodal_projection(const Geometry& g, MultiFab& vel, int vcomp, MultiFab& phi, const MultiFab& sig, int sig_comp,
                         const DomainBC& bc, double rel_tol, double abs_tol, const MGOpts& opts, MultiFab* gp, bool increment_gp,
                         const MultiFab* rhcc = nullptr /* cell-centred source prepared by make_rhcc: div(sig grad phi) = div(vel) + <rhcc> */);
void nodal_rhcc_add(const Geometry& g, MultiFab& rhs, const MultiFab& rc, const DomainBC& bc);
MultiFab make_rhcc(const Geometry& g, const MultiFab& src, int comp, double scale, const MultiFab* drop);
void mask_mult(MultiFab& y, int ycomp, int nc, const MultiFab& m, bool keep_where_zero, int ng);   // y *= (m == 0) or (m != 0), amrns.hip

// Projection::level_project on one level that covers the domain (Source/Projection.cpp:166-450; declaration Projection.H:53-75):
// P_new = 0; U_new /= dt; U_new += Gp_old/rho_half; sigma = 1/rho_half; nodal projection (Gp_new = grad phi, P_new = phi); U_new *= dt.
// U_new: state with the velocity at comps vcomp..vcomp+2, 1 ghost; rho_half: 1 ghost.
MGStats level_project_single(const Geometry& g, double dt, MultiFab& U_new, int vcomp, MultiFab& P_new, const MultiFab& Gp_old, MultiFab& Gp_new,
                             const MultiFab& rho_half, const DomainBC& bc, double proj_tol, double proj_abs_tol, const MGOpts& opts);

// ---- Diffusion (reference Source/Diffusion.H:53-225) -------------------------------------------------
// explicit viscous terms div tau(U): Diffusion::getTensorViscTerms (Source/Diffusion.cpp:1655-1777): out = -b * L_tensor(U), a = 0
// coarse/fine faces of a refined level (tensorop.setCoarseFineBC): crse = the coarse level's velocity (3 comps, valid data on its own
// layout) or null for homogeneous data
struct TensorCF { const MultiFab* crse; const Geometry* cgeom; int ratio; };
// Diffusion::computeExtensiveFluxes after the apply / solve: flux[d] (3 comps, face d) = or += fac * area * operator flux
struct TensorFlux { MultiFab* flux[3]; double fac; bool add; };
void tensor_apply(const Geometry& g, MultiFab& out, MultiFab& vel /*3 comps, 1 ghost; BC data in ghosts*/, double a_scalar, double b_scalar,
                  const MultiFab* acoef, const MultiFab* const eta[3], const DomainBC* bcs, int nbc /*1 or 3 (per component)*/,
                  const TensorCF* cf = nullptr, const TensorFlux* fx = nullptr);
inline void tensor_apply(const Geometry& g, MultiFab& out, MultiFab& vel, double a_scalar, double b_scalar, const MultiFab* acoef,
                         const MultiFab* const eta[3], const DomainBC& bc) { tensor_apply(g, out, vel, a_scalar, b_scalar, acoef, eta, &bc, 1); }
// Crank-Nicolson implicit solve (a*acoef - b div tau) u = rhs: Diffusion::diffuse_tensor_velocity (Source/Diffusion.cpp:837-929)
MGStats tensor_solve(const Geometry& g, MultiFab& soln, const MultiFab& rhs, double a_scalar, double b_scalar, const MultiFab* acoef,
                     const MultiFab* const eta[3], const DomainBC* bcs, int nbc, double tol_rel, double tol_abs, const MGOpts& opts,
                     const TensorCF* cf = nullptr, const TensorFlux* fx = nullptr);
inline MGStats tensor_solve(const Geometry& g, MultiFab& soln, const MultiFab& rhs, double a_scalar, double b_scalar, const MultiFab* acoef,
                            const MultiFab* const eta[3], const DomainBC& bc, double tol_rel, double tol_abs, const MGOpts& opts)
{ return tensor_solve(g, soln, rhs, a_scalar, b_scalar, acoef, eta, &bc, 1, tol_rel, tol_abs, opts); }

// ---- Diffusion operator entries on caller-owned data (diffusion.hip; reference Source/Diffusion.H:53-225) -------------------
// the coarse level's state at the old / new time (valid cells on its own layout; the same component numbering as the fine arrays);
// crse_new == nullptr: homogeneous coarse/fine data (the sync solves)
struct DiffusionCrse { const MultiFab* crse_old; const MultiFab* crse_new; const Geometry* cgeom; int ratio; };
// Diffusion::diffuse_scalar (Source/Diffusion.cpp:207-599): Crank-Nicolson update of component sigma of S_new,
//   (alpha - theta dt div beta grad) s_new = alpha s* + (1 - theta) dt div beta grad s_old + dt delta_rhs,
// rho_flag 0: s = S, alpha = 1; 1: alpha = rho_half; 2: s = S / rho, alpha = rho_new, S_new = s rho_new.  S_old / S_new: at least 1 ghost
// cell, FILLED by the caller (FillPatch: boundary values of the level, Diffusion.cpp:237-239).  fluxn / fluxnp1 (may be null): the
// extensive fluxes (1 - theta) area (-beta grad s_old) and theta area (-beta grad s_new).  add_old_time_divFlux = false: the sync form
// (S_old unused).  Rho_old / Rho_new (null: S_old / S_new): the arrays holding the density in rho_comp.  bc: Diffusion::setDomainBC of the component.
MGStats diffuse_scalar(const Geometry& g, const MultiFab* S_old, const MultiFab* Rho_old, MultiFab& S_new, const MultiFab* Rho_new, int sigma, int rho_comp, double dt, double theta,
                       const MultiFab& rho_half, int rho_flag, MultiFab* const fluxn[3], MultiFab* const fluxnp1[3],
                       const MultiFab* delta_rhs, int rhs_comp, const MultiFab* const betan[3], const MultiFab* const betanp1[3],
                       const DomainBC& bc, const DiffusionCrse* crse, bool add_old_time_divFlux, double visc_tol, const MGOpts& o);
// Diffusion::diffuse_tensor_velocity (Source/Diffusion.cpp:617-957): (alpha - theta dt div tau) u_new = alpha u* + (1 - theta) dt div tau(u_old),
// alpha = rho_half (rho_flag 1) or rho_new with u* weighted by rho_old (rho_flag 3, do_mom_diff).  U_old / U_new: states with the velocity
// in components 0..2 and the density in rho_comp, 1 filled ghost cell.  visc_old_term (may be null): div tau(u_old) if the caller has it
// already and wants no fluxes.  tflux (may be null): the summed extensive viscous fluxes.  fill_new: refills the ghost cells of U_new's
// velocity once it holds rho u* (the FillPatch of Diffusion.cpp:866); null: they are used as they came.
MGStats diffuse_tensor_velocity(const Geometry& g, const MultiFab* U_old, MultiFab& U_new, int rho_comp, double dt, double theta,
                                const MultiFab& rho_half, int rho_flag, const MultiFab* visc_old_term, const MultiFab* const eta_n[3],
                                const MultiFab* const eta_np1[3], const DomainBC bc_visc[3], const DiffusionCrse* crse,
                                MultiFab* const tflux[3], double visc_tol, const MGOpts& o, const std::function<void(MultiFab&)>& fill_new);

// Diffusion::diffuse_tensor_Vsync (Source/Diffusion.cpp:1010-1178) and Diffusion::diffuse_Ssync as NavierStokes::mac_sync uses them; see diffusion.hip
MGStats diffuse_tensor_Vsync(const Geometry& g, MultiFab& Vsync, double dt, double theta, const MultiFab& rho_half, int rho_flag,
                             const MultiFab* Rho_old, const MultiFab* Rho_new, int rho_comp, const MultiFab* const eta[3],
                             const DomainBC bc_visc[3], const BCRec bc_vel[3], const Geometry* cgeom, int ratio, MultiFab* const tflux[3],
                             double visc_tol, const MGOpts& o);
MGStats diffuse_Ssync(const Geometry& g, MultiFab& Ssync, int sn, double dt, double theta, const MultiFab& rho_half, int rho_flag,
                      const MultiFab& Rho_new, int rho_comp, const MultiFab* const beta[3], const DomainBC& bc, const Geometry* cgeom, int ratio,
                      MultiFab* const flux[3], double visc_tol, const MGOpts& o);

// ---- inter-level data motion (amr.hip) ------------------------------------------------------------------------
// amrex::MultiFab::ParallelCopy between different layouts of one index space; periodic_geom != nullptr adds the periodic images
// add = true: dst += src (MultiFab::ParallelAdd); the source regions must then map to disjoint destination cells
class FluxRegister;
// MacProj::mac_sync_solve (Source/MacProj.cpp:359-470)
MGStats mac_sync_solve(const Geometry& g, FluxRegister& mr, const MultiFab& rho_half, double dt, LayoutP fine_layout, int ratio,
                       MultiFab* const Ucorr[3], MultiFab& mac_sync_phi, const DomainBC& bc, double tol, double abs_tol, const MGOpts& opts,
                       const Geometry* cgeom = nullptr, int cratio = 2);

// ---- regrid.hip: error estimation + grid generation
void derive_mag_vort(const Geometry& g, MultiFab& out, int ocomp, const MultiFab& vel, int vcomp);
void error_tag(const Geometry& g, MultiFab& tags, const MultiFab& field, int comp, int mode, double value, int level,
               const double* rb_lo, const double* rb_hi);
// manual_tags_placement at the outflow faces (NavierStokesBase.cpp:2112-2215): mode 1 do_refine_outflow, 2 do_derefine_outflow with
// ncoarse layers (of blocking-factor-coarsened cells) left unrefined
struct OutflowTags { int nface = 0; int dir[6], side[6]; int mode = 0; int ncoarse = 0; };
// periodic-aware erosion of a 0/1 cell map by `passes` cells inside the bounding box [lo, hi] of its set cells (amrregrid.hip: the proper
// nesting domain of a regrid above level 0)
void erode_map(std::vector<unsigned char>& m, const int n[3], const int per[3], int passes, const int lo[3], const int hi[3]);
std::vector<BoxD> cluster_tags(const unsigned char* tags_host, const BoxD& domain, int blocking_factor, int max_grid_size, double grid_eff,
                               int n_error_buf, const OutflowTags* oft = nullptr, const unsigned char* allowed = nullptr /* domain-sized 0/1: where the new level may lie */);

void parallel_copy(MultiFab& dst, const MultiFab& src, int scomp, int dcomp, int nc, int src_ng, int dst_ng, const Geometry* periodic_geom, bool add = false);
// same cells, different boxes (merged working layout <-> the caller's boxes): ghost cells included, valid data last
void relayout_copy(MultiFab& dst, const MultiFab& src, int nc, int scomp = 0, int dcomp = 0);
// the merged layout of a caller's chopped layout if it covers the domain and merging reduces the box count; else null (mf.h: coalesce_layout)
LayoutP merged_solve_layout(const Geometry& g, const LayoutP& l);
// amrex::average_down (cells), average_down_faces, average_down_nodal: NavierStokesBase::avgDown_StatePress, Source/NavierStokesBase.cpp:4125-4193
void average_down(const MultiFab& fine, MultiFab& crse, int scomp, int ncomp, int ratio);
// StateData of one level: old/new MultiFabs and their times (old_ may be null: only one time level)
struct TimeData { const MultiFab* old_; const MultiFab* new_; double t_old, t_new; };
// AmrLevel::FillPatch on a refined level (FillPatchTwoLevels + CellConservativeLinear + physical BC), see amr.hip
// StateData time interpolation (old or new if `time` is within 1e-3 (t_new - t_old) of it); returns the MultiFab to read and its first component
const MultiFab* state_time_interp(const TimeData& td, double time, int scomp, int ncomp, MultiFab& tmp, int& comp0);
void fillpatch_two_levels(MultiFab& dst, int dcomp, double time, const TimeData& fine, const TimeData& crse, int scomp, int ncomp,
                          const Geometry& cgeom, const Geometry& fgeom, int ratio, const BCRec* bc, const double* extdir_lo, const double* extdir_hi);

// NavierStokesBase::create_umac_grown on a refined level: FaceLinear coarse-fine fill of the ghost faces + IAMR's divergence fix
// (Source/NavierStokesBase.cpp:1108-1311); umac_fine need 1 ghost layer, divu (fine cells, >= 1 ghost) may be null
void create_umac_grown(MultiFab* const umac_fine[3], const MultiFab* const umac_crse[3], const MultiFab* divu,
                       const Geometry& cgeom, const Geometry& fgeom, int ratio);
// amrex::FluxRegister / YAFluxRegister role (see amr.hip)
class FluxRegister {
public:
    FluxRegister(LayoutP fine, LayoutP crse, const Geometry& cgeom, int ratio, int ncomp);
    void setVal(double v);
    void CrseInit(const MultiFab& flux /*coarse level, face type dir*/, int dir, int scomp, int dcomp, int nc, double mult, bool add);
    void FineAdd(const MultiFab& flux /*fine level, face type dir*/, int dir, int scomp, int dcomp, int nc, double mult);
    void Reflux(MultiFab& S /*coarse level, cells*/, double volume, double scale, int scomp, int dcomp, int nc);
    const MultiFab& reg(int dir, int side) const { return m_reg[dir][side]; }
private:
    LayoutP m_fine, m_crse;
    Geometry m_cgeom;
    int m_ratio, m_ncomp;
    LayoutP m_slab[3][2];
    MultiFab m_reg[3][2];
};

// ---- multi-level pieces (amrns.hip) -----------------------------------------------------------------------------
// trilinear interpolation (amrex::NodeBilinear) of a coarse nodal MultiFab at the valid nodes of `fine`: fine = (add: +=) interp(crse);
// mask (nodal, fine layout; may be null): only where mask != 0
void node_interp_from_crse(MultiFab& fine, const MultiFab& crse, const Geometry& cgeom, int ratio, const MultiFab* mask, bool add);
// NavierStokesBase::SyncInterp with cell_cons_interp (Source/NavierStokesBase.cpp:3071-3276): conservative-linear interpolation of
// the coarse cell data crse(scomp..) (valid region; periodic images and homogeneous-ext_dir / extrapolated values outside walls are
// built here) to every valid cell of dst(dcomp..)
void sync_interp_cellcons(MultiFab& dst, int dcomp, const MultiFab& crse, int scomp, int ncomp, const Geometry& cgeom, const Geometry& fgeom,
                          int ratio, const BCRec* bc);
// NavierStokesBase::ComputeAofs with is_sync = true (Source/NavierStokesBase.cpp:4594-4845 as called from MacProj::mac_sync_compute):
// edge states traced with umac, fluxes formed with ucorr, conservative update for every component, sync(acomp..) -= update
void godunov_compute_aofs_sync(const Geometry& g, MultiFab& sync, int acomp, const MultiFab& S, int ncomp, const MultiFab* force,
                               const MultiFab* divu, MultiFab* const umac[3], MultiFab* const ucorr[3], const int* iconserv, double dt,
                               const BCRec* bc, bool is_velocity, bool use_forces_in_trans, MultiFab* const flux_out[3], int scheme = 0);

// MacProj::mac_sync_compute on caller-owned arrays (amrns.hip): the form of NavierStokes::mac_sync (Source/MacProj.cpp:488-731) and the form
// with known edge states (:733-786)
void mac_sync_compute(const Geometry& g, MultiFab* const ucorr[3], MultiFab& Vsync, MultiFab& Ssync, const MultiFab& Svel, const MultiFab& Sscal, int nscal,
                      const MultiFab* visc_vel, const MultiFab* tf_scal, const MultiFab& gradp, const MultiFab* divu, MultiFab* const umac[3],
                      const int* iconserv_scal, bool do_mom_diff, double gravity, double dt, const BCRec* bc_vel, const BCRec* bc_scal,
                      bool use_forces_in_trans, int scheme, MultiFab* const flux_vel[3], MultiFab* const flux_scal[3]);
void mac_sync_compute_edge(const Geometry& g, MultiFab* const ucorr[3], MultiFab& Sync, int sync_indx, MultiFab* const edgestate[3], int edge_comp,
                           MultiFab* const flux_out[3]);

// SyncRegister (Source/SyncRegister.cpp): nodal values on the faces of the coarsened fine boxes, kept as ONE single-valued nodal
// MultiFab on the coarse level's layout + the node masks that InitRHS needs
class SyncRegister {
public:
    SyncRegister(LayoutP fine, LayoutP crse, const Geometry& cgeom, const Geometry& fgeom, int ratio, const int phys_lo[3], const int phys_hi[3]);
    void CrseInit(const MultiFab& sync_resid_crse, double mult);          // SyncRegister.cpp:306-318
    void FineAdd(const MultiFab& sync_resid_fine, double mult);           // :350-607
    // :302-348; finer: the boxes of the level above the residual's (ratio finer_ratio to it), fgeom: the residual's level.  Modifies the residual.
    void CompAdd(MultiFab& sync_resid_fine, const Geometry& fgeom, const LayoutP& finer, int finer_ratio, double mult);
    void InitRHS(MultiFab& rhs);                                          // :47-304
    const MultiFab& reg() const { return m_reg; }
    const MultiFab& vs_fine() const { return m_vsfine; }                  // node class of the coarse nodes w.r.t. the fine level: 0 untouched, 1 inside, 2 on its boundary
private:
    LayoutP m_fine, m_crse;
    Geometry m_cgeom, m_fgeom;
    int m_ratio;
    int m_plo[3], m_phi[3];
    MultiFab m_reg, m_onreg, m_vsfine;
};

class NavierStokes;
// sync residual of a level projection (Hydro::NodalProjector::computeSyncResidualCoarse / Fine): crse_side: on the nodes of the level
// that touch both cells covered by the next finer level and cells that are not, rhs - L(phi) formed with the uncovered cells only;
// fine side: on the nodes of the level's own boundary inside the domain, formed with the level's cells only.  Zero elsewhere.
// ---- multi-level nodal projection on caller-owned data (amrns.hip) ----------------------------------------------------------------
// one level of a composite projection: its geometry, boxes, the nodal LinOp BC (Projection.cpp:2432-2464) and the ratio to the next
// coarser level of the solve; gp: the Gradp array that receives (or accumulates) grad phi (may be null); set_inflow: fills the ghost
// velocities outside inflow faces (Projection::set_boundary_velocity), fill_gp: FillPatch of Gradp afterwards (both may be empty)
struct ProjLevel {
    Geometry g; LayoutP layout; DomainBC nodal_bc; int ratio = 2;
    MultiFab* gp = nullptr;
    std::function<void(MultiFab&, double)> set_inflow;
    std::function<void()> fill_gp;
};
// Hydro::NodalProjector::project on the levels PL[0 .. nl-1] (Projection::doMLMGNodalProjection with nlevel > 1, Projection.cpp:2385-2567):
// div(sig grad phi) = div(vel) + rhnd + <rhcc> on the composite grid; vel -= sig grad phi; gp = / += grad phi
MGStats composite_project(const std::vector<ProjLevel>& PL, MultiFab* const vel[], const int vcomp[], MultiFab* const phi[], const MultiFab* const sig[],
                          const MultiFab* rhnd, double rtol, double atol, bool increment_gp, double inflow_scale, const MGOpts& o,
                          const MultiFab* const rhcc[] = nullptr);
// Projection::initialVelocityProject / initialSyncProject (Source/Projection.cpp:615-838, 970-1185) on caller-owned arrays of the levels
// PL[0 .. nl-1] (amrns.hip; the hierarchy's post_init calls the same functions)
MGStats initial_velocity_project(const std::vector<ProjLevel>& PL, MultiFab* const vel[], const int vcomp[], MultiFab* const pres[], const MultiFab* const rho[],
                                 const int rho_comp[], const MultiFab* const divu[], const int divu_comp[], double proj_tol, double proj_abs_tol, const MGOpts& o);
MGStats initial_sync_project(const std::vector<ProjLevel>& PL, MultiFab* const vel_new[], const int vcomp[], const MultiFab* const vel_old[], MultiFab* const phi[],
                             MultiFab* const pres_new[], const MultiFab* const rho_half[], const MultiFab* const divu_new[], const MultiFab* const divu_old[],
                             const int divu_comp[], double dt, double proj_tol, double proj_abs_tol, const MGOpts& o);
// compSyncResidualCoarse (fine_layout given: the level's nodes that touch both cells covered by fine_layout and cells that are not, formed
// with the uncovered cells) / compSyncResidualFine (fine_layout null: the nodes of the level's own boundary inside the domain)
MultiFab sync_resid(const Geometry& g, const LayoutP& layout, const DomainBC& bcn, const LayoutP& fine_layout, int fine_ratio, const MultiFab& vold,
                    const MultiFab& phi, const MultiFab& sig, const MultiFab* rhcc = nullptr);
// Projection::MLsyncProject (Source/Projection.cpp:457-607) on caller-owned data: the two-level sync projection of the velocity increments
// Vsync (coarse level, 3 comps, 1 ghost) and V_corr (fine level: the interpolated Vsync), with sigma = 1 / rho_crse, 1 / rho_fine and the
// sync register's right-hand side; phi_crse / phi_fine (1 ghost, zeroed here) return the pressure correction, which is also added to
// pres_crse / pres_fine; vel_* += dt * the projected increments; PL[l].gp += grad phi.  crse_sync_reg (level PL[0] is itself refined and
// crse_iteration == crse_dt_ratio): receives the residual of the composite solution on PL[0]'s boundary (SyncRegister::CompAdd).
MGStats ml_sync_project(const ProjLevel PL[2], MultiFab& pres_crse, MultiFab& vel_crse, int vcomp_c, MultiFab& pres_fine, MultiFab& vel_fine, int vcomp_f,
                        const MultiFab& rho_crse, const MultiFab& rho_fine, MultiFab& Vsync, MultiFab& V_corr, MultiFab& phi_crse, MultiFab& phi_fine,
                        SyncRegister& rhs_sync_reg, SyncRegister* crse_sync_reg, double dt, int crse_iteration, int crse_dt_ratio,
                        double sync_tol, double abs_tol, const MGOpts& o);
MultiFab amr_sync_resid(NavierStokes& ns, const MultiFab& vold, const MultiFab& phi, const MultiFab& sig, bool crse_side, const MultiFab* rhcc = nullptr /* valid cells */);

// ---- NavierStokes level (reference Source/NavierStokes.cpp:543-691 advance, :1254-1432 post_init) -----
struct NSParams {
    double cfl = 0.8, visc_coef = 0.0, be_cn_theta = 0.5, gravity = 0.0;
    double mac_tol = 1.e-12, mac_abs_tol = 1.e-16, proj_tol = 1.e-12, proj_abs_tol = 1.e-16, visc_tol = 1.e-10;
    int use_forces_in_trans = 0, do_mom_diff = 0, init_iter = 2, init_vel_iter = 1;
    double init_shrink = 1.0, change_max = 1.1, fixed_dt = -1.0;
    int nscal = 2, verbose = 0;
    double init_dt = -1.0;               // ns.init_dt
    double tracer_diff_coef = 0.0;       // ns.scal_diff_coefs[0]
    int phys_lo[3] = {0, 0, 0}, phys_hi[3] = {0, 0, 0};   // ns.lo_bc / ns.hi_bc (PhysBCType: 0 Interior, 4 SlipWall, 5 NoSlipWall)
    double wall_vel_lo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, wall_vel_hi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // xlo.velocity ...: [d*3+n]
    double scal_bc_lo[12] = {0}, scal_bc_hi[12] = {0};   // xlo.density, xlo.tracer, xlo.tracer2, xlo.temp (inflow values): [d*4+n], n = the scalar's slot
    int do_cons_trac = 0;                // ns.do_cons_trac
    int do_denminmax = 0, do_scalminmax = 0;   // ns.do_denminmax / ns.do_scalminmax (NavierStokesBase.cpp:466-467)
    int do_trac2 = 0, do_cons_trac2 = 0; // ns.do_trac2 / ns.do_cons_trac2: a second tracer (NavierStokes.cpp:45-46, NS_setup.cpp:312-320)
    double tracer2_diff_coef = 0.0;      // ns.scal_diff_coefs[1]
    int do_temp = 0;                     // ns.do_temp: temperature, the last state component (NavierStokes.cpp:47-48)
    double temp_cond_coef = 0.0;         // ns.temp_cond_coef
    int use_ppm = 0;                     // ns.advection_scheme: 0 Godunov_PLM, 1 Godunov_PPM (NavierStokesBase.cpp:548-553)
};

enum StateComp { Xvel = 0, Yvel = 1, Zvel = 2, Density = 3, Tracer = 4, MAXSCAL = 4, MAXSTATE = 3 + MAXSCAL, MAXSLOT = MAXSCAL + 2 };   // Tracer2 / Temp: NavierStokes::Tracer2 / Temp (-1: absent)

struct ScalForm { int form[MAXSCAL]; };   // captured by device lambdas
void scale_by(MultiFab& y, const MultiFab& x, int xcomp, int ng, bool divide);   // y (comp 0) *= or /= x(xcomp) on ng ghost cells
class SyncRegister;
class AmrNS;
class NavierStokes {
public:
    // the boxes as the caller (or the grid generator) gave them; `layout` below is coalesce_layout(user_layout): what the level works on
    LayoutP user_layout;
    NavierStokes(const Geometry& g, LayoutP layout, const NSParams& p, const MGOpts& o);
    ~NavierStokes();
    void init_taylorgreen(double vfac, double a, double b, double c, double rho0);   // Source/prob/prob_init.cpp:509-560
    // probtype 10 (Source/prob/prob_init.cpp:407-488, 3-D branch): fluid at rest, tanh density / tracer interface at mid height,
    // perturbed by the hard-coded random amplitude and phases of the reference
    void init_rayleightaylor(double rho_1, double rho_2, double tra_1, double tra_2, double pertamp, double interface_width);
    void init_rest(double rho0);               // probtype 1 (LidDrivenCavity), Source/prob/prob_init.cpp:102-109
    void post_init(double stop_time);          // NavierStokes::post_init
    double step();                             // Amr::coarseTimeStep on one level: computeNewDt + advance
    double advance(double dt) { return advance(dt, 1, 1); }
    double advance(double dt, int iteration, int ncycle);   // NavierStokes::advance(time, dt, iteration, ncycle); returns the dt estimate
    double estTimeStep();                      // NavierStokesBase::estTimeStep
    // derived quantities of the plotfile (derive_lst of NS_setup.cpp:436-449): "energy" = rho |u|^2 / 2 (derkeng), "mag_vort" = |curl u|
    // (dermgvort, ghost cells by FillPatch), "avg_pressure" = mean of the 8 nodes of the cell (deravgpres); new-time data; out: cell, >= 1 comp
    void derive(const std::string& name, MultiFab& out, int ocomp = 0);
    MultiFab& get_new_data(int type) { return type == 0 ? S[inew] : (type == 1 ? P[pnew] : Gp[pnew]); }
    MultiFab& get_old_data(int type) { return type == 0 ? S[1 - inew] : (type == 1 ? P[1 - pnew] : Gp[1 - pnew]); }
    MultiFab& umac(int d) { return u_mac[d]; }
    MultiFab& Aofs() { return aofs; }
    double time = 0.0, dt = 0.0;
    int nstep = 0;
    MGStats st_mac, st_nodal, st_visc, st_scal;
    double t_sections[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // accumulated ms: predict, mac, advect, update, visc, nodal
    bool profile_sections = false;

    // ---- AMR hierarchy (amrns.hip): level index, neighbours, StateData times (State_Type: Point; Press_Type / Gradp_Type: Interval,
    // NavierStokesBase::setTimeLevel, Source/NavierStokesBase.cpp:2978-2996), registers owned by the fine level of an interface
    int level = 0, ratio = 1;
    NavierStokes *crse = nullptr, *fine = nullptr;
    double st_new = 0.0, st_old = 0.0, pt_new[2] = {0.0, 0.0}, pt_old[2] = {0.0, 0.0};
    int iteration = 1, ncycle = 1;
    MultiFab mac_phi;                          // MacProj::mac_phi_crse[level]
    MultiFab rho_avg, p_avg;                   // level > 0
    MultiFab Vsync, Ssync;                     // level < finest
    std::unique_ptr<FluxRegister> reg_adv, reg_visc, reg_mac;
    std::unique_ptr<SyncRegister> sync_reg;
    void set_time_level(double time, double dt_old, double dt_new);
    // checkpoint / restart (AmrLevel::checkPoint / restart role, Source/NavierStokesBase.cpp:856-897, 2706-2727): everything of the level
    // that outlives a time step besides the State / Press / Gradp arrays -- the StateData times, step counters, and the initial-guess
    // history of the MAC solve (this library's addition: it makes a restarted run bit-identical to the uninterrupted one).
    // v[0..15] = time, dt, nstep, st_new, st_old, pt_new[2], pt_old[2], dt_prev_mac, have_mac_prev, have_mac_prev2, dt_min_adv, stop_time, inew, pnew
    void get_restart_state(double v[16]) const;
    void set_stop_time(double t) { m_stop_time = t; }      // amr.restart: stop_time comes from the inputs file
    void set_restart_state(const double v[16]);            // call after the arrays are set; also leaves the initial-step state
    MultiFab& mac_phi_history(int which);                  // 0: last MAC potential, 1: the one before (defined on demand)
    const Geometry& geom() const { return g; }
    const LayoutP& lay() const { return layout; }
    const NSParams& params() const { return p; }
    const DomainBC& nodal_bc() const { return bc_nodal; }
    friend class AmrNS;

private:
    void advance_setup(double dt, int iteration, int ncycle);
    void swap_time_levels(double dt);
    void fill_gp(MultiFab& G, double time);
    void make_rho_curr_time();
    void adv_registers(MultiFab* const flux[3], int state_indx, int ncomp, double dt);
    double predict_velocity(double dt);
    void mac_project(double dt);
    void velocity_advection(double dt);
    void scalar_advection(double dt);
    void scalar_update_rho(double dt);
    void scalar_update_tracers(double dt);
    void scal_min_max(int comp, bool conservative);
    void velocity_advection_update(double dt);
    void scalar_diffusion_update(double dt);
    void get_visc_terms_scalar(MultiFab& visc, MultiFab& Sdata, int comp);
    void scalar_diffusion_update_one(double dt, int sigma);
    void first_order_extrap(MultiFab& mf);
    // data of the coarse level at this level's time t, on the coarse level's own layout (the crsedata of Diffusion.cpp:733-744, 1725-1736)
    void crse_state_at(MultiFab& out, double t, int scomp, int ncomp);
    void crse_scalar_at(MultiFab& out, double t, int comp, bool over_rho);   // a coarse scalar (divided by the coarse density: rho_flag 2)
    double state_time(const MultiFab& Sdata) const { return &Sdata == &S[1 - inew] ? st_old : st_new; }
    const MultiFab& cf_mask();                 // cf_build_mask of the level (2 ghost cells), level > 0
    // div tau(U^n) of the running advance: getViscTerms(prev_time) is asked for by the velocity prediction, by the advection forcing and
    // (times (1 - theta) dt) by the Crank-Nicolson right-hand side -- one tensor apply instead of three (valid only inside advance())
    MultiFab m_visc_old;
    bool m_visc_old_valid = false, m_in_advance = false;
    MultiFab m_cf_mask;
    MultiFab m_mac_phi_prev, m_mac_phi_prev2;  // initial guess of the next MAC solve (last two potentials)
    bool m_have_mac_prev = false, m_have_mac_prev2 = false;
    double dt_prev_mac = 0.0;
    bool m_cf_mask_built = false;
    void fill_gradp_bc();
    void set_inflow_ghosts(MultiFab& vel, double scale);
    bool is_diffusive_scal(int comp) const { return scal_diff[comp - Density] > 0.0; }
    void velocity_diffusion_update(double dt);
    void initial_velocity_diffusion_update(double dt);
    void level_project(double dt);
    void initial_velocity_project();
    void initial_pressure_project();
    bool set_outflow_bcs(MultiFab& phi, const MultiFab& rho, int rcomp);   // hydrostatic data on outflow faces (Projection::set_outflow_bcs)
    LayoutP m_outflow_strip[6];
    double m_stop_time = -1.0;
    void initial_sync_project(double dt);
    void get_visc_terms_vel(MultiFab& visc, MultiFab& Sdata);
    void compute_visc_terms_vel(MultiFab& visc, MultiFab& Sdata);
    const MultiFab& visc_terms_vel_old(MultiFab& scratch);
    const MultiFab& old_visc_or_zero(MultiFab& scratch);
    void advection_all(double dt);
    void fillpatch(MultiFab& dst, const MultiFab& src, int scomp, int ncomp, const BCRec* bc);
    bool is_diffusive_vel() const { return p.visc_coef > 0.0; }

    Geometry g;
    LayoutP layout;
    NSParams p;
    MGOpts o;
    MultiFab S[2], P[2], Gp[2];
    int inew = 0, pnew = 0;
    MultiFab u_mac[3], aofs, rho_ptime, rho_ctime, rho_half;
    MultiFab eta[3];
    double dt_min_adv = 1.e200;
    bool initial_step = false, initial_iter = false;
    // NavierStokes::Initialize (NavierStokes.cpp:43-55): Density, Tracer, [Tracer2], [Temp]; per scalar slot (0 = density): advectionType ==
    // Conservative (NS_setup.cpp:297-320), Diffusion::set_rho_flag(diffusionType) (0 Laplacian_S, 1 RhoInverse_Laplacian_S, 2 Laplacian_SoverRho)
    int nstate = 5, nscal = 2, Tracer2 = -1, Temp = -1;
    // ns.do_temp: Divu_Type and Dsdt_Type exist (NS_setup.cpp:365-383).  They are Point-type cell data with the times of State_Type and are kept
    // here as two more components of the S arrays (Divu = nstate, Dsdt = nstate + 1, nalloc = nstate + 2), with their BCRecs in the slots
    // after the scalars: FillPatch in time and from the coarse level, averaging down and the regrid fill treat them like the state
    bool have_divu = false;
    int nalloc = 5, Divu = -1, Dsdt = -1;
    void calc_divu(bool use_new);                                 // NavierStokes::calc_divu (NavierStokes.cpp:1876-1958)
    void calc_dsdt(double dt);                                    // NavierStokesBase::calc_dsdt (NavierStokesBase.cpp:818-858)
    void divu_half(MultiFab& out, double dt, int ng, bool with_dsdt);   // getDivCond(prev_time) (+ dt/2 getDsdt(prev_time)), zero without divu
    int scal_cons[MAXSCAL] = {1, 0, 0, 0}, scal_rho_flag[MAXSCAL] = {1, 0, 0, 1};
    double scal_diff[MAXSCAL] = {-1.0, 0.0, 0.0, 0.0};
    DomainBC bc_mac, bc_nodal, bc_visc[3], bc_scal_lin[MAXSCAL];
    BCRec bc_vel[3], bc_scal[MAXSLOT], bc_gp[3];
    double ed_vel_lo[9], ed_vel_hi[9];     // ext_dir values [n*3+d]
    double ed_scal_lo[3 * MAXSLOT], ed_scal_hi[3 * MAXSLOT];   // ext_dir (inflow) values of density, tracer, ... [n*3+d]
    MultiFab diff_b[MAXSCAL][3];           // scalar diffusivities on faces (defined for the diffusive slots)
    bool any_wall = false;
};

}  // namespace iamrx
