/*
 * oracle/orc.h -- CPU restatement (plain C, fp64) of IAMR's per-timestep hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liborc.so.  The product (iamr_amd/libiamrx.so)
 * never links, loads or calls anything in this directory.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in AMReX (`AMReX-Codes/amrex`,
 * branch development, unpinned) and AMReX-Hydro (`AMReX-Fluids/AMReX-Hydro`, branch main,
 * unpinned) -- see /root/reference/Exec/Make.IAMR:17-38, Exec/run3d/GNUmakefile:2-3.
 * Neither is present in the reference tree nor in this container, and the reference
 * commits no golden data (Test/README.md:23-29).  This restatement follows the in-tree
 * call sites (cited per function) and the published algorithms (Almgren et al. JCP 142
 * (1998); AMReX MLMG / AMReX-Hydro documentation).  It is pinned against: the exact
 * Taylor vortex (Tutorials/TaylorGreen/benchmarks/EXACT_3D.F:75-119), discrete
 * eigen-answers of the 7-pt and Q1 27-pt operators, and conservation / divergence
 * invariants (tests/test_oracle_*.py).
 *
 * Storage convention = AMReX Array4: for an array allocated on the index
 * region [lo,hi] (inclusive, ghost cells included) with nc components,
 *   offset(i,j,k,n) = (i-lo0) + nx*((j-lo1) + ny*((k-lo2) + nz*n)).
 * Index regions are in the array's own index space: cell (i = cell), x-face (i = face
 * between cells i-1 and i), node (i = lower corner of cell i).
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_fab {
    double* p;
    int lo[3];
    int hi[3];
    int nc;
} orc_fab;

/* geometry of a single-box level: domain cells [0,n-1]^3 */
typedef struct orc_geom {
    int n[3];          /* number of cells */
    double dx[3];
    double problo[3];
    int periodic[3];
} orc_geom;

/* mathematical BC codes = amrex::BCType (AMReX_BC_TYPES.H) */
enum { ORC_BC_REFLECT_ODD = -1, ORC_BC_INT_DIR = 0, ORC_BC_REFLECT_EVEN = 1,
       ORC_BC_FOEXTRAP = 2, ORC_BC_EXT_DIR = 3, ORC_BC_HOEXTRAP = 4 };

/* linear-operator domain BC = amrex::LinOpBCType */
enum { ORC_LO_PERIODIC = 0, ORC_LO_DIRICHLET = 101, ORC_LO_NEUMANN = 102,
       ORC_LO_REFLECT_ODD = 104 /* cell-centred solvers: ghost = -first interior cell (LinOpBCType::reflect_odd, the normal velocity at a Symmetry face) */,
       ORC_LO_INFLOW = 103 /* nodal solver only (LinOpBCType::inflow): Neumann operator, the normal velocity outside the face enters div(u) */ };

/* bcrec: lo[3], hi[3] per component */
typedef struct orc_bcrec { int lo[3]; int hi[3]; } orc_bcrec;

/* ---- ghost filling (orc_fill.c) ------------------------------------------------ */
/* periodic wrap of every ghost entry of `f` whose image lies inside the valid index
 * region; type[d]=1 for nodal direction d.  Mirrors FillBoundary(geom.periodicity()). */
void orc_fill_periodic(orc_fab* f, const orc_geom* g, const int type[3]);
/* physical BC fill of cell-centred ghost cells outside the domain (foextrap, hoextrap,
 * reflect_even/odd, ext_dir with constant value) -- NS_bcfill.H + amrex FilccCell */
/* coarse-fine part of FillPatchTwoLevels with CellConservativeLinear (see orc_fill.c) */
void orc_fill_coarse_fine(orc_fab* fine, const int flo[3], const int fhi[3], const int vlo[3], const int vhi[3],
                          const orc_fab* crse, const int cdomlo[3], const int cdomhi[3], const int periodic[3], int ratio,
                          const orc_bcrec* bc);
/* NavierStokesBase::create_umac_grown on a refined level, single fine box (see orc_fill.c) */
void orc_create_umac_grown(orc_fab* uf[3], const int vlo[3], const int vhi[3], const orc_fab* uc[3], int ratio,
                           const double fdx[3], const int fdom_n[3], const int periodic[3]);
void orc_fill_physbc_cc(orc_fab* f, const orc_geom* g, const orc_bcrec* bc,
                        const double* extdir_lo /*[nc][3]*/, const double* extdir_hi);

/* ---- error estimation (orc_regrid.c) ------------------------------------------------ */
void orc_mag_vort(const orc_geom* g, orc_fab* out, const orc_fab* vel /*3 comps, 1 ghost filled*/);
void orc_error_tag(const orc_geom* g, orc_fab* tags, const orc_fab* f, int comp, int mode, double value, int level,
                   const double* rb_lo, const double* rb_hi);

/* ---- cell-centred ABec operator + multigrid (orc_abec.c) ------------------------ */
typedef struct orc_abec_level {
    orc_geom g;
    double alpha, beta;       /* scalars: (alpha*a - beta div b grad) */
    orc_fab a;                /* cell, 1 comp, 0 ghost (may be p==NULL when alpha==0) */
    orc_fab b[3];             /* face-centred, ncomp comps, 0 ghost */
    int ncomp;
    int tensor;               /* 1: MLTensorOp -- b holds eta*(4/3 on the normal comp), cross terms added in apply */
    int bc_percomp;           /* 1: the lobc/hibc arguments hold ncomp*3 codes, [n*3+d] (MLTensorOp::setDomainBC per component,
                                 reference Source/Diffusion.cpp:724-731) */
    /* AMR level that does not cover the domain (nbox > 0): the level's cells are the union of the boxes (6 ints each: lo, hi);
     * faces of a box that touch neither another box (incl. periodic images) nor the physical boundary are coarse/fine faces with
     * Dirichlet data cf_loc[d] behind them (MLLinOp::setCoarseFineBC: half a coarse cell = 0.5*ratio*dx of the level) */
    int nbox;
    const int* boxes;
    double cf_loc[3];
} orc_abec_level;

typedef struct orc_mg_stats {
    int iters;
    double resnorm0, rhsnorm0, resnorm;
    int bottom_iters_total;
    int converged;
} orc_mg_stats;

typedef struct orc_mg_opts {
    int nu1, nu2, nuf, nub;       /* 2,2,8,0 */
    int max_iters;                /* 200 */
    int bottom_maxiter;           /* 200 */
    double bottom_reltol;         /* 1e-4 */
    double omega;                 /* GSRB over-relaxation (1.15) */
    int maxorder;                 /* Dirichlet ghost extrapolation order */
    int max_coarsening_level;     /* 30 */
    int min_width;                /* coarsest box width (2) */
    int nodal_sweeps;             /* nodal: GS sweeps per smooth call (4) */
    int nodal_smoother;           /* 0 = 8-colour GS, 1 = lexicographic GS, 2 = Jacobi(2/3) */
    int verbose;
    int bottom_smoother_only;     /* 1: bottom = nuf smooths instead of BiCGStab */
    int fixed_iters;              /* >0: do exactly this many V-cycles, ignore tolerances */
} orc_mg_opts;

void orc_mg_default_opts(orc_mg_opts* o);
extern int orc_threads;
void orc_set_threads(int n);      /* OpenMP threads of the smoother colour loops (default 1) */

void orc_abec_apply(const orc_abec_level* L, orc_fab* y, const orc_fab* x /*1 ghost, filled*/);
void orc_abec_gsrb(const orc_abec_level* L, orc_fab* phi /*1 ghost, filled*/, const orc_fab* rhs,
                   int redblack, double omega, const int lobc[3], const int hibc[3], int maxorder);
void orc_abec_applybc(const orc_abec_level* L, orc_fab* phi, const int lobc[3], const int hibc[3],
                      int maxorder, int inhomog, const orc_fab* bcval /*same shape as phi, ghost cells hold BC values*/);
/* Dirichlet data of the coarse/fine faces of level L (nbox > 0): the coarse solution cphi (cell, 1 ghost, ncomp comps; periodic ghosts
 * filled by the caller) interpolated to the fine ghost cells -- InterpBndryData::setBndryValues, third order in the tangential
 * directions (quadratic with one-sided / dropped terms where the neighbouring ghost cell is not a coarse/fine ghost cell).
 * bcval: cell, 1 ghost, 3*ncomp comps, [n*3+d] = value for a face of direction d */
void orc_cf_interp_bndry(const orc_abec_level* L, int ratio, const orc_fab* cphi, orc_fab* bcval);
/* orc_abec_solve on a level with coarse/fine faces: cf_bcval from orc_cf_interp_bndry (NULL: homogeneous) */
void orc_abec_solve_cf(const orc_abec_level* L, orc_fab* phi, const orc_fab* rhs, const int lobc[3], const int hibc[3],
                       const orc_fab* cf_bcval, double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);
/* MacProj::mac_project on an AMR level > 0 (Source/MacProj.cpp:225-353 with cphi = mac_phi_crse[level-1], :1166-1170): the level is the
 * union of nbox boxes (6 ints each) of the fine index space g; cphi = coarse-level MAC phi (cell, 1 ghost, periodic ghosts filled).
 * umac / rho / S / phi are whole-domain fabs of the fine index space, only the entries of the level are used / changed; rho needs its
 * ghost cells next to the level filled (FillPatch from the coarse level). */
void orc_mac_project_cf(const orc_geom* g, orc_fab* umac[3], const orc_fab* rho, const orc_fab* S, orc_fab* phi, double rhs_scale,
                        const int lobc[3], const int hibc[3], int nbox, const int* boxes, int ratio, const orc_fab* cphi,
                        double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);
void orc_cc_restrict(orc_fab* crse, const orc_fab* fine, const int cn[3]);
void orc_cc_prolong_add(orc_fab* fine, const orc_fab* crse, const int fn[3]);
void orc_face_avgdown(orc_fab* crse, const orc_fab* fine, int dir, const int cn[3]);

/* solve (alpha*a - beta div b grad) phi = rhs on the single-box level.  phi has 1 ghost
 * (ghost values at Dirichlet faces = inhomogeneous BC data on entry).  Follows AMReX
 * MLMG::solve/oneIter/mgVcycle/actualBottomSolve + MLCGSolver::solve_bicgstab. */
void orc_abec_solve(const orc_abec_level* L, orc_fab* phi, const orc_fab* rhs,
                    const int lobc[3], const int hibc[3],
                    double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);

/* flux_d = -beta * b_d * dphi/dx_d on faces (MLMG::getFluxes) */
void orc_abec_flux(const orc_abec_level* L, orc_fab* flux[3], const orc_fab* phi);

/* Hydro::MacProjector::project as called from MacProj::mlmg_mac_solve
 * (reference Source/MacProj.cpp:1084-1184): b = 1/(rhs_scale*rho_face),
 * rhs = S - div(umac), solve -div(b grad phi) = rhs, umac -= b grad phi. */
void orc_mac_project(const orc_geom* g, orc_fab* umac[3] /*faces, >=0 ghost*/, const orc_fab* rho /*cell, 1 ghost filled*/,
                     const orc_fab* S /*cell, may be NULL*/, orc_fab* phi /*cell, 1 ghost*/,
                     double rhs_scale, const int lobc[3], const int hibc[3],
                     double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);

void orc_mac_divergence(const orc_geom* g, orc_fab* div, orc_fab* const umac[3]);

/* ---- Godunov (orc_godunov.c) ------------------------------------------------------ */
double orc_slope4(const orc_fab* q, int i, int j, int k, int n, int dir);
/* Godunov::ExtrapVelToFaces (PLM), reference call site Source/NavierStokesBase.cpp:4487-4491 */
void orc_extrap_vel_to_faces(const orc_geom* g, const orc_fab* vel /*3 comps, 3 ghost filled*/,
                             const orc_fab* force /*3 comps, 1 ghost*/, orc_fab* umac[3],
                             double dt, const orc_bcrec* bc /*[3]*/, int use_forces_in_trans);
/* HydroUtils::ComputeFluxesOnBoxFromState("Godunov", PLM) + ComputeDivergence(mult=-1) +
 * ComputeConvectiveTerm, then aofs = -update; reference Source/NavierStokesBase.cpp:4701-4842 */
void orc_compute_aofs(const orc_geom* g, orc_fab* aofs /*ncomp comps starting at acomp*/, int acomp,
                      const orc_fab* S /*ncomp comps, 3 ghost*/, int ncomp,
                      const orc_fab* force /*ncomp,1 ghost*/, const orc_fab* divu /*1 ghost or NULL*/,
                      orc_fab* const umac[3], const int* iconserv, double dt,
                      const orc_bcrec* bc, int is_velocity, int use_forces_in_trans,
                      orc_fab* edge_out[3] /*optional, ncomp face comps*/, orc_fab* flux_out[3]);

/* edge-state reconstruction used by the two routines above: 0 PLM (default), 1 PPM */
void orc_godunov_set_ppm(int scheme);   /* ns.advection_scheme: 0 Godunov_PLM, 1 Godunov_PPM, 2 BDS */
int orc_godunov_get_ppm(void);

/* ---- nodal projection (orc_nodal.c) ---------------------------------------------- */
void orc_nodal_adotx(const orc_geom* g, orc_fab* y, const orc_fab* x /*node,1 ghost*/, const orc_fab* sig /*cell,1 ghost*/);
void orc_nodal_divu(const orc_geom* g, orc_fab* rhs /*node*/, const orc_fab* vel /*cell,3 comps,1 ghost*/);
void orc_nodal_smooth(const orc_geom* g, orc_fab* x, const orc_fab* rhs, const orc_fab* sig, int smoother, int nsweeps,
                      const int lobc[3], const int hibc[3]);
void orc_nodal_restrict(orc_fab* crse, const orc_fab* fine, const orc_geom* cg);
void orc_nodal_interp_add(orc_fab* fine, const orc_fab* crse, const orc_fab* sig_fine, const orc_geom* fg);
void orc_nodal_mknewu(const orc_geom* g, orc_fab* vel, const orc_fab* phi, const orc_fab* sig);
void orc_nodal_compgrad(const orc_geom* g, orc_fab* gp, const orc_fab* phi);
void orc_nodal_solve(const orc_geom* g, orc_fab* phi, const orc_fab* rhs, const orc_fab* sig,
                     const int lobc[3], const int hibc[3], double rtol, double atol,
                     const orc_mg_opts* o, orc_mg_stats* st);
/* Hydro::NodalProjector::project as called from Projection::doMLMGNodalProjection
 * (reference Source/Projection.cpp:2512-2542): rhs = div(vel), solve div(sig grad phi)=rhs,
 * vel -= sig grad phi. */
/* the same solve on a level that covers only the cells with cov != 0 (NULL: all) and/or has Dirichlet (outflow) domain faces:
 * nodes on the boundary of the covered region keep the incoming phi (MLNodeLaplacian Dirichlet mask) */
void orc_nodal_solve_cov(const orc_geom* g, orc_fab* phi, const orc_fab* rhs, const orc_fab* sig,
                         const int lobc[3], const int hibc[3], const orc_fab* cov /*cell, 0 ghost*/, double rtol, double atol,
                         const orc_mg_opts* o, orc_mg_stats* st);
void orc_nodal_project_cov(const orc_geom* g, orc_fab* vel, orc_fab* phi, const orc_fab* sig, const int lobc[3], const int hibc[3],
                           const orc_fab* cov, double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);
void orc_nodal_rhcc_add(const orc_geom* g, orc_fab* rhs, const orc_fab* rhcc, const int lobc[3], const int hibc[3], const orc_fab* cov);
void orc_nodal_project_rhcc(const orc_geom* g, orc_fab* vel, orc_fab* phi, const orc_fab* sig, const int lobc[3], const int hibc[3],
                            const orc_fab* cov, const orc_fab* rhcc, double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);
void orc_nodal_project(const orc_geom* g, orc_fab* vel /*3 comps, 1 ghost*/, orc_fab* phi /*node, 1 ghost*/,
                       const orc_fab* sig, const int lobc[3], const int hibc[3],
                       double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);

/* ---- tensor diffusion (orc_tensor.c) ---------------------------------------------- */
/* y = (alpha*a - beta div tau(u)) with MLTensorOp semantics (3 comps) */
void orc_tensor_apply(const orc_geom* g, orc_fab* y, const orc_fab* u /*3 comps,1 ghost filled incl. corners*/,
                      double alpha, double beta, const orc_fab* a, orc_fab* const eta[3] /*faces,1 comp*/);
void orc_tensor_apply_bcn(const orc_geom* g, orc_fab* y, orc_fab* u /*3 comps, 1 ghost: BC data in, operator ghosts out*/,
                          double alpha, double beta, const orc_fab* a, orc_fab* const eta[3],
                          const int* lobc /*9: [n*3+d]*/, const int* hibc, int maxorder);
void orc_tensor_solve_bcn(const orc_geom* g, orc_fab* u, const orc_fab* rhs, double alpha, double beta,
                          const orc_fab* a, orc_fab* const eta[3], const int* lobc /*9: [n*3+d]*/, const int* hibc,
                          double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);
void orc_tensor_solve(const orc_geom* g, orc_fab* u, const orc_fab* rhs, double alpha, double beta,
                      const orc_fab* a, orc_fab* const eta[3], const int lobc[3], const int hibc[3],
                      double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);

/* ---- level time step (orc_ns.c) ---------------------------------------------------- */
typedef struct orc_ns_params {
    double cfl, visc_coef, be_cn_theta;
    double gravity;
    double mac_tol, mac_abs_tol, proj_tol, proj_abs_tol, visc_tol;
    int use_forces_in_trans;
    int do_mom_diff;
    int init_iter, init_vel_iter;
    double init_shrink, change_max, fixed_dt;
    int nscal;                 /* number of scalars incl. density (2) */
    int verbose;
    double init_dt;            /* ns.init_dt: used when estTimeStep finds no velocity/force scale (-1: abort) */
    double tracer_diff_coef;   /* ns.scal_diff_coefs[0]: tracer diffusivity (<= 0: not diffusive) */
    int phys_lo[3], phys_hi[3];/* ns.lo_bc / ns.hi_bc: 0 Interior (periodic), 1 Inflow, 2 Outflow, 3 Symmetry, 4 SlipWall, 5 NoSlipWall (Source/NS_BC.H) */
    double wall_vel_lo[9], wall_vel_hi[9]; /* xlo.velocity ... zhi.velocity: [d*3+n] = comp n on the lo/hi face of direction d */
    double scal_bc_lo[12], scal_bc_hi[12]; /* xlo.density, xlo.tracer, xlo.tracer2, xlo.temp (inflow values): [d*4+n], n = the scalar's slot (0 density, 1 tracer, then tracer2 / temp as present) */
    int do_cons_trac;          /* ns.do_cons_trac (Source/NS_setup.cpp:306-310): Conservative advection, Laplacian_SoverRho diffusion */
    int do_denminmax, do_scalminmax;   /* ns.do_denminmax / ns.do_scalminmax (Source/NavierStokesBase.cpp:466-467, 2771-2788, 2907-2935) */
    int do_trac2, do_cons_trac2;       /* ns.do_trac2 / ns.do_cons_trac2: a second tracer (NavierStokes.cpp:45-46, NS_setup.cpp:312-320) */
    double tracer2_diff_coef;          /* ns.scal_diff_coefs[1] */
    int do_temp;                       /* ns.do_temp: temperature as the last state component (NavierStokes.cpp:47-48), Divu_Type / Dsdt_Type exist */
    double temp_cond_coef;             /* ns.temp_cond_coef */
    int use_ppm;               /* ns.advection_scheme = Godunov_PPM (Source/NavierStokesBase.cpp:548-553); 0: Godunov_PLM */
} orc_ns_params;

typedef struct orc_ns_state orc_ns_state;

void orc_ns_default_params(orc_ns_params* p);
orc_ns_state* orc_ns_create(const orc_geom* g, const orc_ns_params* p, const orc_mg_opts* o);
void orc_ns_destroy(orc_ns_state* s);
/* pointers to the persistent arrays (for initial conditions and comparison).
 * which: 0 S_new (nstate comps, 1 ghost), 1 S_old, 2 P_new (node,1 ghost), 3 P_old,
 * 4 Gp_new (3 comps,1 ghost), 5 Gp_old, 6..8 umac, 9 aofs */
orc_fab* orc_ns_fab(orc_ns_state* s, int which);
void orc_ns_init_taylorgreen(orc_ns_state* s, double vfac, double a, double b, double c, double rho0);
/* prob.probtype = 10 (Source/prob/prob_init.cpp:407-488, 3-D branch) */
void orc_ns_init_rayleightaylor(orc_ns_state* s, double rho_1, double rho_2, double tra_1, double tra_2, double pertamp, double interface_width);
/* NavierStokes::post_init sequence: initialVelocityProject, estimate dt, init_iter pressure iterations */
void orc_ns_init_rest(orc_ns_state* s, double rho0);     /* probtype 1: LidDrivenCavity start */
void orc_ns_test_set_extrap_scale(double v);            /* test hook, see orc_ns.c first_order_extrap */
void orc_ns_post_init(orc_ns_state* s, double stop_time);
/* one coarse time step: computeNewDt + NavierStokes::advance; returns dt used */
double orc_ns_step(orc_ns_state* s);
double orc_ns_time(const orc_ns_state* s);
double orc_ns_dt(const orc_ns_state* s);
void orc_ns_last_stats(const orc_ns_state* s, orc_mg_stats* mac, orc_mg_stats* nodal, orc_mg_stats* visc);

/* ---- multi-level hierarchy (orc_amr.c): Amr::coarseTimeStep with subcycling, reflux, average down, MAC sync, sync projection ------- */
/* every level > 0 is the union of nbox[l] boxes (6 ints each: lo, hi, in the level's own index space, aligned to the refinement
 * ratio), held as whole-domain arrays of that index space: orc_ns_fab(orc_amr_level(a, l), which) are fabs on [0, n0*ratio^l - 1]^3;
 * only the entries inside the boxes are data of the level.  boxes: concatenation for levels 1 .. nlev-1; nbox[0] is ignored. */
typedef struct orc_amr orc_amr;
orc_amr* orc_amr_create(const orc_geom* g0, const orc_ns_params* p, const orc_mg_opts* o, int nlev, int ratio, const int* nbox, const int* boxes);
void orc_amr_destroy(orc_amr* a);
orc_ns_state* orc_amr_level(orc_amr* a, int lev);
const orc_fab* orc_amr_cov(orc_amr* a, int lev);            /* cell fab, 1 on the level's cells (NULL on level 0) */
/* NavierStokes::post_init for the hierarchy; S_new of every level must hold the initial data on the level's cells */
void orc_amr_post_init(orc_amr* a, double stop_time);
double orc_amr_coarse_step(orc_amr* a);                      /* returns the level-0 dt */
double orc_amr_time(const orc_amr* a);
double orc_amr_dt(const orc_amr* a, int lev);
void orc_amr_sync_stats(const orc_amr* a, orc_mg_stats* st);
/* Hydro::NodalProjector::project on levels c0 .. c0+nl-1 (composite nodal projection), see orc_amr.c */
void amr_composite_project(orc_amr* a, int c0, int nl, orc_fab* vel[], orc_fab* phi[], const orc_fab* sig[], const orc_fab* rhnd, double rtol,
                           double atol, int increment_gp, double inflow_scale, orc_mg_stats* st);
/* ComputeAofs with is_sync (orc_godunov.c) */
void orc_compute_aofs_sync(const orc_geom* g, orc_fab* sync, int acomp, const orc_fab* S, int ncomp,
                           const orc_fab* force, const orc_fab* divu, orc_fab* const umac[3], orc_fab* const ucorr[3], const int* iconserv,
                           double dt, const orc_bcrec* bc, int is_velocity, int use_forces_in_trans, orc_fab* flux_out[3]);

#ifdef __cplusplus
}
#endif
#endif
