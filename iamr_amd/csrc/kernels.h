// iamr_amd/csrc/kernels.h -- host-callable launchers of the hand-written HIP kernels (gfx950).
#pragma once
#include "core.h"
#include "mf.h"

namespace iamrx {

// ---- k_basic.hip --------------------------------------------------------------------------
void launch_fill(double* p, size_t n, double v, hipStream_t s);
void launch_copy_plan(const CopyDesc* d, int nd, long maxpts, const FabD* src, const FabD* dst, int scomp, int dcomp, int nc, hipStream_t s, bool add = false);
void launch_pack(const CopyDesc* d, int nd, long maxpts, const FabD* src, double* buf, long pts_total, int scomp, int nc, hipStream_t s);
void launch_unpack(const CopyDesc* d, int nd, long maxpts, const FabD* dst, const double* buf, long pts_total, int dcomp, int nc, hipStream_t s, bool add = false);
// the same three over a flat work list (mf.h CopyWork: nw entries (descriptor, chunk)): one workgroup per COPY_CHUNK points that exist
void launch_copy_plan_w(const CopyDesc* d, const int2* w, int nw, const FabD* src, const FabD* dst, int scomp, int dcomp, int nc, hipStream_t s, bool add);
void launch_pack_w(const CopyDesc* d, const int2* w, int nw, const FabD* src, double* buf, long pts_total, int scomp, int nc, hipStream_t s);
void launch_unpack_w(const CopyDesc* d, const int2* w, int nw, const FabD* dst, const double* buf, long pts_total, int dcomp, int nc, hipStream_t s, bool add);
// global: combined over the ranks (on the device, before the single read-back) unless the layout is replicated
double reduce_norm0(const MultiFab& mf, int comp, int nc, int ng, bool global = false);
void reduce_norm0_comps(const MultiFab& mf, int comp, int nc, int ng, double* out, bool global = false);   // per-component maxima, one read-back
void reduce_minmax(const MultiFab& mf, int comp, int ng, double& mn, double& mx, bool global = true);   // one pass, one read-back
double reduce_sum_unique(const MultiFab& mf, int comp, const Geometry& g, bool global = false);   // sum over owner copies
// nout simultaneous dot products over the valid region, owner-masked for nodal data: out[q] = <x_q, y_q>
void reduce_dots(int nout, const MultiFab* const* x, const MultiFab* const* y, int comp, int nc, const Geometry& g, double* out, bool local = false);
// ... with the results left on the device (no read-back): krylov.h
void reduce_dots_dev(int nout, const MultiFab* const* x, const MultiFab* const* y, int comp, int nc, const Geometry& g, double* d_out, bool local = false);
// y = a*x + b*y etc. (valid region + ng)
// HIP-event probe around the k_nodal_gs4 launches of levels with >= min_nodes nodes per box (see k_nodal.hip)
// HIP-event probes around the launches of one kernel family on levels with at least min_points cells / nodes per box (every stride-th one)
enum { PROBE_NODAL_GS4 = 0, PROBE_ABEC_GSRB = 1, PROBE_GOD_Z = 2, PROBE_PRED_Z = 3, PROBE_COUNT = 4 };
void kernel_probe_start(int which, long min_points, int stride);
void kernel_probe_stop(int which, double* total_ms, long* launches);
void kernel_probes_pause(bool on);      // no probe events while a launch sequence is being captured into a graph
bool kernel_probe_begin(int which, long points);     // true: the start event was recorded, call kernel_probe_end after the launch
void kernel_probe_end(int which);
void gs4_probe_start(long min_nodes, int stride);
void gs4_probe_stop(double* total_ms, long* launches);
void mf_lincomb(MultiFab& dst, double a, const MultiFab& x, double b, const MultiFab& y, int comp, int nc, int ng);   // dst = a*x + b*y
void mf_saxpy(MultiFab& y, double a, const MultiFab& x, int xcomp, int ycomp, int nc, int ng);                          // y += a*x
void mf_add_scalar(MultiFab& y, double a, int comp, int nc, int ng);
void mf_mult(MultiFab& y, double a, int comp, int nc, int ng);

struct DomainBC;
// ---- k_bc.hip -----------------------------------------------------------------------------
// physical-BC fill of cell-centred ghost cells outside the domain; extdir_lo/hi[n*3+d] constant ext_dir values
void fill_physbc_cc(const Geometry& g, MultiFab& mf, int scomp, int ncomp, const BCRec* bc, const double* extdir_lo, const double* extdir_hi);

void nodal_reflect_bc(const Geometry& g, MultiFab& mf, const DomainBC& bc, hipStream_t on = nullptr);   // ghost nodes: even reflection about Neumann walls
void cc_mirror_bc(const Geometry& g, MultiFab& mf);                           // cell-centred mirror across all non-periodic walls

// ---- k_abec.hip ---------------------------------------------------------------------------
struct AbecCoef {
    double alpha, beta;
    const MultiFab* a;        // cell, 1 comp (may be null)
    const MultiFab* b[3];     // face, ncomp comps (or 1 comp broadcast if b_ncomp == 1)
    int tensor;               // add MLTensorOp cross terms in apply/residual
    int tensor_eta = 0;       // b[d] hold the 1-component face viscosity eta_d; the kernels apply b_d(comp) = eta_d * (comp == d ? 4/3 : 1)
    // sig != null: b[d] was made by mac_bcoef(b, *sig, sig_comp, sig_scale) and the smoother / residual kernels may recompute the face
    // value sig_scale / (0.5 (sig(cell - e_d) + sig(cell))) from the cell-centred array instead of reading three face arrays (the same
    // expression, hence the same doubles; one array of HBM traffic instead of three).  sig has >= 1 ghost cell, filled.
    const MultiFab* sig = nullptr;
    int sig_comp = 0;
    double sig_scale = 1.0;
    // b_uniform: every entry of the one-component b[d] equals bu[d] (constant viscosity / diffusivity: mf_uniform_value found it so);
    // the smoother and residual kernels use the constants instead of reading the three face arrays
    int b_uniform = 0;
    double bu[3] = {0.0, 0.0, 0.0};
};
// every valid entry of component 0 of m equals one value (on all ranks): returns true and the value
bool mf_uniform_value(const MultiFab& m, double* v);
// the tensor operator's residual / apply in ONE launch where its viscosity is constant (k_tensor.hip); false: use abec_residual
bool tensor_residual_fused(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& vel, const MultiFab* rhs, double* norm_out);
struct DomainBC {             // linear-operator BC of the level's domain
    int lo[3], hi[3];         // LinOpBC per face
    int maxorder;
};
// Coarse/fine faces of a level that does not cover the domain (MLLinOp::setCoarseFineBC): Dirichlet data half a coarse cell behind
// the face.  c[d][NX-2][m]: Lagrange weights of the ghost formula through x = {-loc/dx, 0.5, 1.5, 2.5} (m = 0: the coarse datum),
// NX = min(box length + 1, maxorder).
struct CfTab { double c[3][3][4]; int maxorder; };
CfTab cf_make_tab(const double loc[3], const double dx[3], int maxorder);
// cell mask, ghost cells included: 0 = cell of the level (valid, neighbour box or periodic image), 1 = coarse/fine ghost cell,
// 2 = outside the physical domain
void cf_build_mask(const Geometry& g, MultiFab& cfm);
// ghost cells with mask 1 next to a box face: phi = c[0] * bcval (inhomog) + sum_m c[m] * phi(m-th cell inside)   (mllinop_apply_bc)
// edges: also the edge / corner coarse-fine ghost cells (tensor operator): bcval there (cf_interp_edges) or zero
void cf_fill_ghosts(MultiFab& phi, const MultiFab& cfm, const CfTab& tab, bool inhomog, const MultiFab* bcval, bool edges = false);
void cf_interp_edges(MultiFab& bcval, const MultiFab& cpatch, const MultiFab& cfm, int ratio, const Geometry& cgeom);
// bcval(ghost cells with mask 1) = coarse data of cpatch (coarsened layout, 1 ghost cell) interpolated in the tangential directions
// (InterpBndryData::setBndryValues, third order, ratio 2); cfm needs 2 ghost cells
void cf_interp_bndry(MultiFab& bcval, const MultiFab& cpatch, const MultiFab& cfm, int ratio);
// bcs: nbc DomainBC entries (nbc == 1: same BC for all components; nbc == ncomp: one per component, MLTensorOp::setDomainBC)
void abec_gsrb(const Geometry& g, const AbecCoef& c, MultiFab& phi, const MultiFab& rhs, int redblack, double omega, const DomainBC* bcs, int nbc,
               bool shell_only = false, bool wrap = false, const MultiFab* cfm = nullptr, const CfTab* cftab = nullptr, bool cf_maintain_ghosts = false,
               bool phi_is_zero = false, bool walls_inkernel = false);
// walls_inkernel: the pass applies the homogeneous domain boundary conditions itself (no ghost cell of phi is read in a non-periodic direction:
// the caller fills periodic ghost cells only) -- allowed where this returns true
bool abec_gsrb_walls_inkernel_ok(const Geometry& g, const AbecCoef& c, const MultiFab& phi, int nbc, const DomainBC* bcs, bool cf);
// the last two levels of a cell-centred V-cycle in one single-workgroup launch (k_abec_tail, k_abec.hip): pre-smoothing from zero, residual,
// restriction, bottom solve, prolongation, post-smoothing -- the doubles of the launches it replaces
bool abec_tail_ok(const Geometry& gF, const Layout& lF, const Geometry& gC, const Layout& lC, const AbecCoef& cF, const DomainBC* bcs, int nbc, int ncomp);
void abec_tail_solve(const Geometry& gF, const AbecCoef& cF, MultiFab& corF, const MultiFab& resF, const Geometry& gC, const AbecCoef& cC,
                     const DomainBC& bc, bool singular, double eps_rel, int maxiter, int nub, int nuf, int nu1, int nu2, double omega, int* d_iters);
// phi_is_zero: the pass may be told that phi is identically zero (the first pass on a multigrid correction) INSTEAD of phi being set to
// zero in front of it -- it then reads no phi and writes every cell (the active colour its update, the other colour zero) -- if this returns
// true for the same arguments (one component, one-component coefficients, one box spanning a periodic domain: no ghost cell is read)
// restriction of the residual rhs - A phi straight onto the coarsened layout (one pass, the fine residual is not stored): usable if ..._ok
bool abec_residual_reads_no_ghosts(const Geometry& g, const AbecCoef& c, const MultiFab& out, const MultiFab& phi, const MultiFab& rhs, bool restrict_form);
bool abec_resid_restrict_ok(const AbecCoef& c, const MultiFab& phi, const MultiFab& rhs);
void abec_resid_restrict(const Geometry& g, const AbecCoef& c, MultiFab& crse, const MultiFab& phi, const MultiFab& rhs);
bool abec_gsrb_zero_ok(const AbecCoef& c, const MultiFab& phi, int nbc, bool wrap, bool has_cf);
// one red + black sweep in ONE launch, out of place (pin -> pout), on a level that is one box spanning a periodic domain (k_abec_gsrb_rb: the
// doubles of the two colour passes, a third of their HBM traffic); zero: pin is identically zero and is not read
bool abec_gsrb_rb_ok(const Geometry& g, const AbecCoef& c, const MultiFab& phi, int nbc, const DomainBC* bcs = nullptr);
// cf: the level is a refined box strictly inside its domain (abec_gsrb_rb_cf_ok): its coarse/fine ghost formula, evaluated inside the kernel
void abec_gsrb_rb(const Geometry& g, const AbecCoef& c, const MultiFab& pin, MultiFab& pout, const MultiFab& rhs, double omega, bool zero,
                  const DomainBC* bcs = nullptr, int nbc = 0, const CfTab* cf = nullptr, bool acc = false);
// (acc: pout is the SOLUTION of the running solve and receives pout + the swept correction -- the last sweep of a V-cycle, k_abec.hip ACC)
bool abec_gsrb_rb_cf_ok(const Geometry& g, const AbecCoef& c, const MultiFab& phi);
// the same sweep on a level of several boxes that covers its domain (a chopped level, the boxes of a sharded level): k_abec_rb_ghost +
// k_abec_gsrb_rb<.., NBR>, one two-layer ghost fill of phi per sweep in front of it (the caller's).  level_ok: the layout / boundary
// conditions admit it (the caller then gives phi two ghost layers, rhs and the a-term one, the density two); ok: these arrays do
bool abec_gsrb_rb_nbr_level_ok(const Geometry& g, const Layout& l, int ncomp, bool sig_form, bool has_a, int nbc, const DomainBC* bcs);
bool abec_gsrb_rb_nbr_ok(const Geometry& g, const AbecCoef& c, const MultiFab& phi, const MultiFab& rhs, int nbc, const DomainBC* bcs);
// sel 0: the whole sweep; 1: only the tiles that read no ghost cell (no k_abec_rb_ghost launch); 2: k_abec_rb_ghost + the other tiles;
// on: the stream (null: the context's).  splits: parts 1 and 2 are both non-empty (and the level has no ghost columns in x)
void abec_gsrb_rb_nbr(const Geometry& g, const AbecCoef& c, MultiFab& pin, MultiFab& pout, const MultiFab& rhs, double omega, bool zero,
                      const DomainBC* bcs, int nbc, int sel = 0, hipStream_t on = nullptr, bool acc = false);
bool abec_gsrb_rb_nbr_splits(const Geometry& g, const Layout& l);
// fused red+black sweep, out of place; see k_abec.hip (the caller refreshes the ghosts of phi_out and finishes the black cells
// on box surfaces with abec_gsrb(..., 1, ..., shell_only = true))
void abec_gsrb_fused(const Geometry& g, const AbecCoef& c, const MultiFab& phi_in, MultiFab& phi_out, const MultiFab& rhs, double omega,
                     const DomainBC* bcs, int nbc);
// out = rhs - L(phi)  (rhs == nullptr: out = L(phi))
void abec_residual(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& phi, const MultiFab* rhs, double* norm_out = nullptr);
// bottom solve (BiCGStab + post-smoothing, CellMG::bottom_solve) of a single-box level of at most 8^3 cells in one single-workgroup launch
bool abec_bottom_device_ok(const Geometry& g, const Layout& l, const DomainBC* bcs, int nbc, int ncomp, bool cf = false);
void abec_bottom_solve(const Geometry& g, const AbecCoef& c, MultiFab& cor, const MultiFab& res, const DomainBC& bc, bool singular,
                       double eps_rel, int maxiter, int nub, int nuf, double omega, int* d_iters, const CfTab* cftab = nullptr);
void abec_apply_domain_bc(const Geometry& g, MultiFab& phi, const DomainBC& bc, bool inhomog, const MultiFab* bcval, int comp0 = 0, int ncomp = -1);
void abec_apply_domain_bc_percomp(const Geometry& g, MultiFab& phi, const DomainBC* bcs, int ncomp, bool inhomog, const MultiFab* bcval);
void cc_restrict(MultiFab& crse, const MultiFab& fine);          // average of 8
void cc_prolong_add(MultiFab& fine, const MultiFab& crse);       // piecewise constant
void face_avgdown(MultiFab& crse, const MultiFab& fine, int dir);
// flux_d = -beta*b_d*dphi/dx_d ; if add_to != nullptr: add_to[d] += flux_d instead of storing
void abec_flux(const Geometry& g, const AbecCoef& c, const MultiFab& phi, MultiFab* const flux[3], MultiFab* const add_to[3]);
void mac_rhs(const Geometry& g, MultiFab& rhs, const MultiFab* const umac[3], const MultiFab* S);   // rhs = S - div(umac)
void mac_bcoef(MultiFab* const b[3], const MultiFab& rho, int rho_comp, double scale);                 // b = scale / avg_face(rho)
void mac_divergence(const Geometry& g, MultiFab& div, const MultiFab* const umac[3]);

// ---- k_godunov.hip ------------------------------------------------------------------------
// Godunov::ExtrapVelToFaces (PLM): vel has >=3 comps and >=3 filled ghost cells, force 3 comps >=1 ghost
// scheme: ns.advection_scheme, 0 Godunov_PLM (4th-order limited slopes), 1 Godunov_PPM -- an argument of every call, no process-wide mode
void godunov_extrap_vel_to_faces(const Geometry& g, const MultiFab& vel, const MultiFab* force, MultiFab* const umac[3],
                                 double dt, const BCRec* bc, bool use_forces_in_trans, int scheme = 0);
// ComputeFluxesOnBoxFromState + ComputeDivergence(-1) + ComputeConvectiveTerm, aofs(acomp..) = -update.
// S: ncomp comps, >=3 ghosts; umac: >=1 ghost (filled); force/divu: >=1 ghost or null
void godunov_compute_aofs(const Geometry& g, MultiFab& aofs, int acomp, const MultiFab& S, int ncomp, const MultiFab* force,
                          const MultiFab* divu, MultiFab* const umac[3], const int* iconserv, double dt, const BCRec* bc,
                          bool is_velocity, bool use_forces_in_trans, MultiFab* const edge_out[3], MultiFab* const flux_out[3], int scheme = 0);


// ---- k_nodal.hip --------------------------------------------------------------------------
bool nodal_residual(const Geometry& g, MultiFab& out, const MultiFab& x, const MultiFab& sig, const MultiFab* rhs, double* norm_out = nullptr);
void nodal_gs_color(const Geometry& g, MultiFab& x, const MultiFab& rhs, const MultiFab& sig, int color, const MultiFab* dmask = nullptr);
// one k-parity pass of the plane-fused 8-colour GS (arrays need ngrow >= 4 / 3), out of place: plane k from xc, planes
// k+-1 from xn, result to xo (xo != xc; xn may be either)
void nodal_gs_fused_pass(const Geometry& g, const MultiFab& xc, const MultiFab& xn, MultiFab& xo, const MultiFab& rhs, const MultiFab& sig, int kpar,
                         bool wrap = false, const MultiFab* dmask = nullptr, const double* csig = nullptr, int zero_flags = 0, int refl = 0,
                         int sel = 0, hipStream_t on = nullptr);
// sel (k_nodal_gsr only): 1 = the tiles that read no ghost node of x (footprint and planes inside the box), 2 = the others, 0 = all; on: the
// stream of the launch (null: the context's).  nodal_gsr_splits: the level has tiles of the first kind
bool nodal_gsr_splits(const MultiFab& x, const MultiFab& rhs, const MultiFab* dmask);
// (wrap: one box spanning its domain, images instead of ghost nodes -- periodic ones, or mirror images in the directions of refl (bit d):
// nodal_wrap_or_reflect_ok)
// k_nodal_gsr takes the level: then zero_flags (bit 0: xc, bit 1: xn is identically zero and is not read) may be passed
bool nodal_gsr_applies(const MultiFab& x, const MultiFab& rhs, const MultiFab* dmask);
void nodal_zero_masked(MultiFab& mf, const MultiFab& dmask);
void nodal_build_dmask(const Geometry& g, MultiFab& dm, const MultiFab& cov, const DomainBC& bc);
bool periodic_wrap_ok(const Geometry& g, const Layout& l, int min_len);
bool nodal_wrap_or_reflect_ok(const Geometry& g, const Layout& l, const DomainBC& bc, int min_len, int* refl);
// all sweeps x 8 colours of a small single-box periodic level in one single-workgroup launch (false: not applicable)
bool nodal_smooth_small(const Geometry& g, MultiFab& x, const MultiFab& rhs, const MultiFab& sig, int nsweeps);
void nodal_jacobi(const Geometry& g, MultiFab& xnew, const MultiFab& x, const MultiFab& rhs, const MultiFab& sig, const MultiFab* dmask = nullptr);
void nodal_restrict(MultiFab& crse, const MultiFab& fine);
void nodal_interp_add(MultiFab& fine, const MultiFab& crse, const MultiFab& sig_fine);
void nodal_divu(const Geometry& g, MultiFab& rhs, const MultiFab& vel, int vcomp, const DomainBC* bc);
// vel(vcomp..) -= sig*grad(phi) (vel may be null); gp (may be null) = or += grad(phi)
void nodal_mknewu(const Geometry& g, MultiFab* vel, int vcomp, const MultiFab& phi, const MultiFab* sig, MultiFab* gp, bool gp_increment);

// ---- k_tensor.hip -------------------------------------------------------------------------
void tensor_bcoef(MultiFab& b3, const MultiFab& eta, int dir);
// extensive face fluxes of the tensor operator (Diffusion::computeExtensiveFluxes): fac * area * (-eta (4/3) du_n/dx_d + cross terms)
void tensor_extensive_flux(const Geometry& g, const MultiFab& vel, const MultiFab* const eta[3], MultiFab* const flux[3], double fac, bool add);
void tensor_cross_terms_sub(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& vel, double sign, unsigned long long* normout = nullptr);
void fill_tensor_corners(const Geometry& g, MultiFab& phi, const DomainBC& bc, bool inhomog, const MultiFab* bcval, int comp0 = 0, int ncomp = -1);

}  // namespace iamrx
