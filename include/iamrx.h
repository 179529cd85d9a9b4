/*
 * include/iamrx.h -- C-ABI of libiamrx.so: the MI355X-native (gfx950) replacement for the
 * data-parallel hot path of AMReX-Fluids/IAMR.
 *
 * Conventions
 *   - every entry point returns 0 on success, non-zero on failure; iamrx_last_error() gives the
 *     message.  IAMR itself has no return codes (errors are amrex::Abort, e.g. reference
 *     Source/NavierStokesBase.cpp:501-507): the reference-side wrapper turns non-zero into Abort.
 *   - arrays are double precision, AMReX Array4 layout seen at IAMR's seam (reference
 *     Source/NavierStokesBase.cpp:4665-4677): offset(i,j,k,n) = (i-lo0) + n0*((j-lo1) + n1*((k-lo2) + n2*n)),
 *     ghost cells included; index boxes are inclusive [lo,hi].
 *   - all compute runs on the library's HIP stream of the selected device; there is NO CPU fallback:
 *     iamrx_init fails if no gfx950 device is present.
 *   - handles are opaque; objects are owned by the library until the matching *_destroy.
 *
 * Each function cites the reference interface it replaces (file:line relative to /root/reference).
 */
#ifndef IAMRX_H
#define IAMRX_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iamrx_layout_s* iamrx_layout;   /* BoxArray + DistributionMapping of one level */
typedef struct iamrx_fluxreg_s* iamrx_fluxreg; /* amrex::FluxRegister of one coarse/fine interface (see below) */
typedef struct iamrx_mf_s* iamrx_mf;           /* MultiFab (device resident) */
typedef struct iamrx_ns_s* iamrx_ns;           /* NavierStokes level object */
typedef struct iamrx_syncreg_s* iamrx_syncreg; /* SyncRegister of one coarse/fine interface */
typedef struct iamrx_amr_s* iamrx_amr;         /* hierarchy of NavierStokes levels (the Amr / AmrLevel role for the hot path) */

typedef struct iamrx_geom {
    int dom_lo[3], dom_hi[3];       /* cell-centred domain box */
    double prob_lo[3], prob_hi[3];
    int periodic[3];
} iamrx_geom;

/* MLMG controls (defaults = reference defaults, except the nodal cycle shape, see nodal_nu1: Source/MacProj.cpp:41-101, Source/Projection.cpp:19-37,
 * Source/Diffusion.cpp:85-96 and upstream amrex::MLMG nu1=nu2=2, nuf=8, bottom BiCGStab rtol 1e-4) */
typedef struct iamrx_mg_opts {
    int nu1, nu2, nuf, nub;
    int max_iters, bottom_maxiter;
    double bottom_reltol;
    double omega;
    int maxorder;
    int max_coarsening_level, min_width;
    int nodal_sweeps, nodal_smoother;
    int verbose;
    int bottom_smoother_only;
    int fixed_iters;
    /* nodal V-cycle shape: pre / post smooth calls (nodal_sweeps Gauss-Seidel sweeps each).  amrex::MLNodeLaplacian / MLMG use 4 sweeps and
     * 2 + 2 calls; iamrx_mg_default_opts() gives 2 sweeps and 1 + 1 calls -- the converged solution is the same, the time to solution is
     * 1/3 shorter on MI355X.  Set nodal_sweeps = 4, nodal_nu1 = nodal_nu2 = 2 for the upstream cycle. */
    int nodal_nu1, nodal_nu2;
    /* cell-centred bottom: 1 (default) = the hierarchy ends at the first single-box level of <= 8^3 cells, solved by one single-workgroup
     * device launch (BiCGStab to bottom_reltol + nub sweeps, no host synchronisation); 0 = coarsen to min_width, host-driven BiCGStab
     * (amrex::MLMG's shape).  Same converged solution. */
    int device_bottom;
    /* 1: the level is a 2-D problem lifted onto a thin periodic slab (iamr_amd/inputs.py lift_2d): once the slab is two cells thick the
     * multigrid keeps it at two cells and coarsens the plane only.  0 (default): ordinary coarsening.  A property
     * of the problem handed to each solver, not process-wide state (the IAMRX_MG_SLAB switch still forces it for tests). */
    int slab;
} iamrx_mg_opts;

typedef struct iamrx_mg_stats {
    int iters;
    double resnorm0, rhsnorm0, resnorm;
    int bottom_iters_total;
    int converged;             /* 1: the tolerance was met; 2 (nodal solves): the residual stalled at the fp64 round-off floor within 100x of the tolerance (h <= 1/512) */
    double vcycle_ms;
    int nlevels;
} iamrx_mg_stats;

/* ---- runtime ---------------------------------------------------------------------------- */
int iamrx_init(int device);                      /* amrex::Initialize role (Source/main.cpp:26-40) */
int iamrx_finalize(void);
const char* iamrx_last_error(void);
int iamrx_sync(void);                            /* amrex::Gpu::synchronize */
void* iamrx_stream(void);                        /* the hipStream_t every kernel is launched on */
int iamrx_mem_info(size_t* bytes_live, size_t* bytes_cached);
int iamrx_alloc_count(size_t* n_device_malloc);  /* hipMalloc calls so far (caching-allocator misses; The_Arena role) */
/* Run-time switches (kernel variants kept for A/B measurement and as references of the parity tests, solver
 * shortcuts, tile shapes) live in ONE registry.  It is filled once, at iamrx_init, from the environment variables IAMRX_<KEY> that are
 * set (e.g. IAMRX_GODUNOV_Z=0 -> key "GODUNOV_Z"), can be changed afterwards only through iamrx_tuning_set, and is consulted at the
 * point of use -- a change takes effect at the next call; nothing is cached for the life of the process.  Per-solve choices that belong
 * to a call are arguments (iamrx_mg_opts, the Godunov `scheme`), not keys. */
int iamrx_tuning_set(const char* key, double value);
int iamrx_tuning_get(const char* key, double default_value, double* value);
/* scoped wall-time profile of the library's host-side sections (measurement aid; synchronises the stream at every scope boundary while
 * enabled).  Writes the accumulated report (one line per scope path: name, ms, calls) into report[capacity] first (may be NULL), then
 * enable: 1 on, 0 off, -1 unchanged; reset != 0 clears the accumulated times. */
int iamrx_scope_profile(int enable, int reset, char* report, size_t capacity);
/* halo exchanges with other ranks since iamrx_init: out[0] exchanges issued on the main stream (in front of the kernel that needs the data:
 * exposed), out[1] doubles sent by them; out[2] / out[3] the same for exchanges issued on the side stream beside interior work (the
 * multi-box red + black sweep: hidden as far as the interior tiles last); the role of the comm rows of a TINY_PROFILE */
int iamrx_exchange_counts(size_t out[4]);
/* how often the level objects have merged a caller's boxes so far (IAMRX_COALESCE = 1, the default): lets a test
 * harness tell the calls whose result depends on that mode (tests/conftest.py runs those in both).  No counterpart upstream. */
int iamrx_coalesce_merge_count(size_t* n);
int iamrx_sync_count(size_t* n_stream_sync);     /* host waits on the library stream so far (scalar read-backs of norms / dot products, plan uploads) */
/* HIP-event stopwatch on the library stream (the role of BL_PROFILE / ParallelDescriptor::second() pairs,
 * e.g. Source/NavierStokesBase.cpp:2088-2107): start records an event, stop records + waits and returns ms */
int iamrx_timer_start(void);
int iamrx_timer_stop(double* ms);
void iamrx_mg_default_opts(iamrx_mg_opts* o);
/* HIP-event probes around the launches of one kernel family inside running solves (the role of TINY_PROFILE rows such as MLMG smoother
 * times): which = 0 nodal Gauss-Seidel pass (k_nodal_gs4), 1 scalar GSRB colour pass (k_abec_gsrb), 2 fused Godunov advection (k_god_z), 3 fused
 * velocity prediction (k_pred_z); only launches on levels
 * with at least min_points nodes / cells per box, every stride-th one.  stop waits for the stream and returns the summed durations. */
int iamrx_kernel_probe_start(int which, long min_points, int stride);
int iamrx_kernel_probe_stop(int which, double* total_ms, long* launches);

/* ---- communicator (amrex::ParallelDescriptor / FabArray point-to-point role) ------ */
/* One process per GPU.  Call AFTER iamrx_init and BEFORE creating layouts.  RCCL: rank 0 obtains a unique id,
 * the host program broadcasts the 128 bytes (MPI_Bcast / torch.distributed), every rank calls init. */
int iamrx_comm_get_unique_id(char id[128]);
int iamrx_comm_init_rccl(const char id[128], int rank, int nranks);
/* transport supplied by the host program (tests: torch.distributed gloo); buffers are host memory.
 * op: 0 sum, 1 max, 2 min */
typedef void (*iamrx_allreduce_cb)(double* vals, int n, int op);
typedef void (*iamrx_exchange_cb)(int nsend, const int* send_peers, double** send_bufs, const long* send_counts,
                                  int nrecv, const int* recv_peers, double** recv_bufs, const long* recv_counts);
int iamrx_comm_init_callback(int rank, int nranks, iamrx_allreduce_cb ar, iamrx_exchange_cb ex);
int iamrx_comm_rank(int* rank, int* nranks);
/* transport probe: exchange `count` doubles with `peer` through the installed communicator (peer == own rank: loop-back); 0 = data intact */
int iamrx_comm_probe_exchange(int peer, long count);
const char* iamrx_comm_last_error(void);

/* ---- containers (amrex::BoxArray/DistributionMapping/MultiFab role) ----------- */
int iamrx_layout_create(int nboxes, const int* lo_hi /* 6 ints per box: lo[3],hi[3] */, const int* owner_rank,
                        iamrx_layout* out);
/* The level objects (iamrx_ns_*, iamrx_amr_*) work on the caller's boxes MERGED: boxes of one owner rank that share a full face become one box,
 * repeatedly (one process drives one GPU, so small boxes inside a rank only cost ghost fills and launches; the reference's amr.max_grid_size
 * chopping -- RunningProblems.rst:362-368 -- is undone inside the level and restored at its data accessors, which keep speaking the caller's
 * boxes).  This returns the merged boxes of a layout; IAMRX_COALESCE = 0 (iamrx_tuning_set("COALESCE", 0)) switches the merging off. */
int iamrx_layout_coalesced_boxes(iamrx_layout l, int* nboxes, int* lo_hi /* 6 ints per box, or NULL to query the count */);
int iamrx_layout_destroy(iamrx_layout l);
int iamrx_layout_nlocal(iamrx_layout l, int* nlocal);
int iamrx_layout_local_box(iamrx_layout l, int local_idx, int lo_hi[6], int* global_idx);
int iamrx_mf_create(iamrx_layout l, const int type[3] /* 1 = nodal in that direction */, int ncomp, int ngrow, iamrx_mf* out);
int iamrx_mf_destroy(iamrx_mf m);
int iamrx_mf_info(iamrx_mf m, int* ncomp, int* ngrow, int type[3], int* nlocal);
int iamrx_mf_fab_box(iamrx_mf m, int local_idx, int lo_hi[6]);         /* allocated region incl. ghosts */
int iamrx_mf_dev_ptr(iamrx_mf m, int local_idx, double** p);
int iamrx_mf_to_host(iamrx_mf m, int local_idx, double* dst);            /* whole fab, all comps */
int iamrx_mf_from_host(iamrx_mf m, int local_idx, const double* src);
int iamrx_mf_setval(iamrx_mf m, double v);                                /* MultiFab::setVal */
int iamrx_mf_copy(iamrx_mf dst, iamrx_mf src, int scomp, int dcomp, int ncomp, int ngrow);  /* MultiFab::Copy */
int iamrx_mf_fill_boundary(iamrx_mf m, const iamrx_geom* g);              /* FillBoundary(geom.periodicity()): Source/MacProj.cpp:1127 */
/* physical-BC fill of cell-centred ghost cells outside the domain (the BCRec part of FillPatch; ext_dir with constant
 * boundary values as in Source/NS_bcfill.H:17-95 / Source/NavierStokes.cpp:72-83).  bcrec: [ncomp][6];
 * extdir_lo/hi: [ncomp][3] boundary values (may be NULL) */
int iamrx_mf_fill_physbc(iamrx_mf m, const iamrx_geom* g, int scomp, int ncomp, const int* bcrec, const double* extdir_lo,
                         const double* extdir_hi);
int iamrx_mf_norm0(iamrx_mf m, int comp, int ncomp, int ngrow, double* out);   /* MultiFab::norm0: Source/NavierStokesBase.cpp:4408 */

/* host-only (works without a GPU): the ghost-exchange plan that `rank` executes for FillBoundary of a level
 * (FabArray::FillBoundary role).  desc: 16 ints per descriptor = kind (0 local copy, 1 pack+send, 2 recv+unpack),
 * peer rank, src global box, dst global box, region lo[3] hi[3] (destination index frame), shift[3]
 * (src index = dst index + shift), buffer offset in points, 2 pad.  Call with desc == NULL to size. */
int iamrx_host_fill_plan(int nboxes, const int* lo_hi, const int* owner, int rank, const int type[3], int ngrow, const iamrx_geom* g,
                         int max_desc, int* desc, int* ndesc);
/* the same for a fill whose SOURCE boxes also hand on their first `wall_ext` ghost layers beyond the non-periodic sides of the domain
 * (cell-centred data; the two-layer density copy of the multi-box sweep: the edge ghost cells beyond a wall behind a box-box face take
 * the face ghost values of the box next door).  No counterpart upstream: amrex::FillBoundary copies valid cells only. */
int iamrx_host_fill_plan_wall_ext(int nboxes, const int* lo_hi, const int* owner, int rank, const int type[3], int ngrow, const iamrx_geom* g,
                                  int wall_ext, int max_desc, int* desc, int* ndesc);

/* ---- cell-centred linear operator primitives (amrex::MLABecLaplacian role) ---- */
/* one red or black Gauss-Seidel pass of (alpha*a - beta div b grad) phi = rhs; ghost cells of phi must be filled */
int iamrx_abec_gsrb(const iamrx_geom* g, double alpha, double beta, iamrx_mf a /* may be NULL */, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                    iamrx_mf phi, iamrx_mf rhs, int redblack, double omega, const int lobc[3], const int hibc[3], int maxorder);
/* one full red+black sweep with the ghost/BC fill in front of each colour (homogeneous BC, as inside MLMG::mgVcycle);
 * fused = 1: single-pass kernel (red everywhere + black off the box surfaces) followed by the black pass on the box surfaces */
/* The finest-level kernel forms of the cell-centred multigrid on caller data (inside a solve CellMG picks them by itself): coef = 1: the MAC
 * operator with its face coefficients b_d = scale / (0.5 (rho(cell - e_d) + rho(cell))) recomputed from the cell-centred rho (>= 1 filled ghost
 * cell; the role of MacProj.cpp:1098-1128's coefficients), coef = 2: uniform face coefficients bu[3] (constant viscosity / diffusivity).
 * op 0 / 1: red / black colour pass on phi (ghost cells filled by the caller; abec_gsrb role); 4 / 5: the same with the index wrap of one box
 * spanning a periodic domain (no ghost cells read); 2: out = rhs - L phi; 3: out (on the layout coarsened by 2) = restriction of rhs - L phi;
 * 6: out = phi after one red + black sweep in ONE launch (one box spanning the domain, 128 or 256 cells in x, every side periodic, Neumann,
 * reflect-odd or Dirichlet of order <= 3 -- homogeneous, as inside a V-cycle; no ghost cell of phi is read; out != phi, >= 1 ghost cell),
 * 7: the same with phi taken as zero without being read; 8 / 9: the sweep of 6 / 7 on a level of SEVERAL boxes that covers its domain (a
 * chopped level; the boxes of a level sharded over GPUs; rows of 128 / 256 cells in every box): phi and out with two ghost layers, rhs
 * with one, rho with two -- the entry fills them from the neighbour boxes / periodic images (one exchange per sweep instead of one per
 * colour; rho's ghost cells beyond domain walls stay the caller's), updates the red ghost cells next to box faces in place and sweeps;
 * 10 / 11: the sweep of 6 / 7 on a REFINED level that is one box strictly inside the domain g (ratio 2 to the level below; the role of
 * MLLinOp::setCoarseFineBC's homogeneous ghost values inside MLMG::mgVcycle on a level set up by MacProj.cpp:1166-1170): every face of the
 * box is a coarse/fine face whose ghost value of order `maxorder` (<= 4: the face cell, the cell behind it and the one behind that) the
 * kernel forms itself; no ghost cell of phi is read, rho's first ghost layer is the caller's (FillPatch). */
int iamrx_abec_form(const iamrx_geom* g, int coef, iamrx_mf rho, int rho_comp, double scale, const double bu[3], double beta, int op,
                    iamrx_mf phi, iamrx_mf rhs, iamrx_mf out, double omega, const int lobc[3], const int hibc[3], int maxorder);
int iamrx_abec_gsrb_sweep(const iamrx_geom* g, double alpha, double beta, iamrx_mf a /* may be NULL */, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                          iamrx_mf phi, iamrx_mf rhs, double omega, const int lobc[3], const int hibc[3], int maxorder, int fused);
/* out = rhs - L(phi)  (rhs == NULL: out = L(phi)) */
int iamrx_abec_residual(const iamrx_geom* g, double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                        iamrx_mf out, iamrx_mf phi, iamrx_mf rhs, int tensor);
int iamrx_cc_restrict(iamrx_mf crse, iamrx_mf fine);
int iamrx_cc_prolong_add(iamrx_mf fine, iamrx_mf crse);
int iamrx_face_avgdown(iamrx_mf crse, iamrx_mf fine, int dir);
/* full MLMG solve: amrex::MLMG::solve on MLABecLaplacian (tensor = 1: MLTensorOp, b = eta faces with 1 comp) */
int iamrx_abec_solve(const iamrx_geom* g, double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                     iamrx_mf phi, iamrx_mf rhs, const int lobc[3], const int hibc[3], double rtol, double atol,
                     const iamrx_mg_opts* o, int tensor, iamrx_mg_stats* st);

/* ---- MAC projection -------------------------------------------------------------------- */
/* MacProj::mlmg_mac_solve (Source/MacProj.cpp:1084-1184; declaration Source/MacProj.H:82-94):
 * b = (1/rhs_scale)/rho_face, rhs = S - div(u_mac), solve, u_mac -= b grad phi.  rho needs 1 filled ghost. */
int iamrx_mlmg_mac_solve(const iamrx_geom* g, iamrx_mf umac_x, iamrx_mf umac_y, iamrx_mf umac_z, iamrx_mf rho, int rho_comp,
                         iamrx_mf S /* may be NULL */, iamrx_mf mac_phi, double rhs_scale, const int lobc[3], const int hibc[3],
                         double mac_tol, double mac_abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);
/* the same on an AMR level > 0: cphi = mac_phi_crse[level-1] (the coarse level's MAC phi on its own layout, geometry cgeom) supplies the
 * Dirichlet data of the coarse/fine faces (macproj.setCoarseFineBC(cphi, ratio), Source/MacProj.cpp:1166-1170).  rho needs its ghost cells
 * filled (FillPatch from the coarse level at coarse/fine faces). */
int iamrx_mlmg_mac_solve_cf(const iamrx_geom* g, iamrx_mf umac_x, iamrx_mf umac_y, iamrx_mf umac_z, iamrx_mf rho, int rho_comp,
                            iamrx_mf S /* may be NULL */, iamrx_mf mac_phi, double rhs_scale, const int lobc[3], const int hibc[3],
                            iamrx_mf crse_phi, const iamrx_geom* cgeom, int ratio, double mac_tol, double mac_abs_tol,
                            const iamrx_mg_opts* o, iamrx_mg_stats* st);
/* MacProj::mac_sync_solve (Source/MacProj.cpp:359-470): coarse-level solve for the correction velocity that removes the mismatch between
 * the coarse MAC velocity and the fine one on the coarse/fine interface.  mac_reg: the interface's register after
 * CrseInit(u_mac_crse, mult = -area_crse) and FineAdd(u_mac_fine, mult = +area_fine / ncycle) (Source/MacProj.cpp:306-346);
 * ucorr_*: coarse face arrays (out), mac_sync_phi: coarse cells, 1 ghost (out); fine: the fine level's layout. */
int iamrx_mac_sync_solve(const iamrx_geom* g, iamrx_fluxreg mac_reg, iamrx_mf rho_half, double dt, iamrx_layout fine, int ratio,
                         iamrx_mf ucorr_x, iamrx_mf ucorr_y, iamrx_mf ucorr_z, iamrx_mf mac_sync_phi, const int lobc[3], const int hibc[3],
                         double mac_sync_tol, double mac_abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);
/* MacProj::check_div_cond (Source/MacProj.cpp:792-846): div = div(u_mac) */
int iamrx_mac_divergence(const iamrx_geom* g, iamrx_mf div, iamrx_mf umac_x, iamrx_mf umac_y, iamrx_mf umac_z);

/* ---- Godunov advection ---------------------------------------------------------------- */
/* bcrec: 6 ints per component (lo[3], hi[3]) = amrex::BCRec mathematical BC codes
 * (reference Source/NS_BC.H:7-55). */
/* scheme (last argument of the three entries): the edge-state reconstruction, ns.advection_scheme / godunov_use_ppm
 * (Source/NavierStokesBase.cpp:548-553, 4654-4656): IAMRX_GODUNOV_PLM 0, IAMRX_GODUNOV_PPM 1.  Per call: the library keeps no
 * process-wide advection mode. */
#define IAMRX_GODUNOV_PLM 0
#define IAMRX_GODUNOV_PPM 1
#define IAMRX_BDS 2          /* Bell-Dawson-Shubin edge states in ComputeAofs (the velocity prediction stays Godunov_PLM, as upstream) */
/* Godunov::ExtrapVelToFaces as called from NavierStokesBase::predict_velocity
 * (Source/NavierStokesBase.cpp:4487-4491): vel (3 comps, 3 filled ghosts), force (3 comps, 1 ghost) -> u_mac */
int iamrx_godunov_extrap_vel_to_faces(const iamrx_geom* g, iamrx_mf vel, iamrx_mf force, iamrx_mf umac_x, iamrx_mf umac_y,
                                      iamrx_mf umac_z, double dt, const int* bcrec /* [3][6] */, int use_forces_in_trans, int scheme);
/* the kernel chain of NavierStokesBase::ComputeAofs (Source/NavierStokesBase.cpp:4594-4845):
 * ComputeFluxesOnBoxFromState("Godunov") + ComputeDivergence(mult=-1, area weighted) + ComputeConvectiveTerm,
 * aofs(acomp+n) = -update.  edge_* / flux_* (ncomp face comps) are optional outputs (NULL to skip). */
int iamrx_godunov_compute_aofs(const iamrx_geom* g, iamrx_mf aofs, int acomp, iamrx_mf S, int ncomp, iamrx_mf force, iamrx_mf divu,
                               iamrx_mf umac_x, iamrx_mf umac_y, iamrx_mf umac_z, const int* iconserv, double dt,
                               const int* bcrec /* [ncomp][6] */, int is_velocity, int use_forces_in_trans,
                               iamrx_mf edge_x, iamrx_mf edge_y, iamrx_mf edge_z, iamrx_mf flux_x, iamrx_mf flux_y, iamrx_mf flux_z, int scheme);

/* ---- nodal projection (amrex::MLNodeLaplacian / Hydro::NodalProjector role) ------- */
/* out = rhs - div(sig grad phi) at nodes (rhs == NULL: out = div(sig grad phi)); phi and sig need 1 filled ghost */
int iamrx_nodal_residual(const iamrx_geom* g, iamrx_mf out, iamrx_mf phi, iamrx_mf sig, iamrx_mf rhs);
/* one colour (0..7) of the 8-colour Gauss-Seidel sweep */
int iamrx_nodal_gs_color(const iamrx_geom* g, iamrx_mf phi, iamrx_mf rhs, iamrx_mf sig, int color);
/* one full 8-colour Gauss-Seidel sweep incl. its ghost fills.  fused = 0: eight colour passes (8 fills);
 * fused = 1: plane-fused variant, two passes (2 fills), identical arithmetic; needs phi/sig ngrow >= 4, rhs >= 3
 * (rhs ghosts must be filled by the caller) */
int iamrx_nodal_gs_sweep(const iamrx_geom* g, iamrx_mf phi, iamrx_mf rhs, iamrx_mf sig, int fused);
int iamrx_nodal_restrict(iamrx_mf crse, iamrx_mf fine);
int iamrx_nodal_interp_add(iamrx_mf fine, iamrx_mf crse, iamrx_mf sig_fine);
int iamrx_nodal_divu(const iamrx_geom* g, iamrx_mf rhs, iamrx_mf vel, int vcomp);
/* MLNodeLaplacian::compGrad as used by NavierStokesBase::computeGradP (Source/NavierStokesBase.cpp:4102-4122) */
int iamrx_nodal_compgrad(const iamrx_geom* g, iamrx_mf gp, iamrx_mf phi);
/* Projection::doMLMGNodalProjection, one level (Source/Projection.cpp:2385-2567; declaration Source/Projection.H:244-254):
 * rhs = div(vel), solve div(sig grad phi) = rhs, vel -= sig grad phi, gp = (increment_gp ? += : =) grad phi */
int iamrx_nodal_projection(const iamrx_geom* g, iamrx_mf vel, int vcomp, iamrx_mf phi, iamrx_mf sig, int sig_comp,
                           const int lobc[3], const int hibc[3], double rel_tol, double abs_tol, const iamrx_mg_opts* o,
                           iamrx_mf gp /* may be NULL */, int increment_gp, iamrx_mg_stats* st);

/* iamrx_abec_solve on an AMR level > 0 (MLLinOp::setCoarseFineBC + setLevelBC as MacProj::mlmg_mac_solve and the Diffusion solves use
 * them on refined levels, Source/MacProj.cpp:1166-1170): the level of phi does not cover the domain; its box faces inside the domain
 * that touch no other box of the level are coarse/fine faces with Dirichlet data taken from crse_phi (the coarse level's solution on its
 * own layout, geometry cgeom; NULL: zero), interpolated along the face (InterpBndryData, third order) and applied with the level's
 * maxorder half a coarse cell behind the face. */
int iamrx_abec_solve_cf(const iamrx_geom* g, double alpha, double beta, iamrx_mf a, iamrx_mf bx, iamrx_mf by, iamrx_mf bz,
                        iamrx_mf phi, iamrx_mf rhs, const int lobc[3], const int hibc[3], iamrx_mf crse_phi, const iamrx_geom* cgeom,
                        int ratio, double rel_tol, double abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);

/* the nodal solve alone: div(sig grad phi) = rhs on the level `phi` lives on (MLMG::solve on MLNodeLaplacian, as driven by
 * Hydro::NodalProjector inside Projection::doMLMGNodalProjection, Source/Projection.cpp:2512-2542, and by the sync solves of
 * Projection::MLsyncProject, :457-607).  LinOpBC codes: Neumann (walls, inflow), Dirichlet (outflow).  Nodes on Dirichlet domain
 * faces and -- if the level does not cover the domain (an AMR level > 0) -- on the level's boundary inside the domain are
 * Dirichlet nodes: they keep the incoming phi (MLNodeLaplacian's Dirichlet mask). */
int iamrx_nodal_solve(const iamrx_geom* g, iamrx_mf phi, iamrx_mf rhs, iamrx_mf sig, int sig_comp, const int lobc[3], const int hibc[3],
                      double rel_tol, double abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);

/* ---- tensor diffusion (amrex::MLTensorOp role) -------------------------------------- */
/* out = (a*acoef - b div tau(vel)); Diffusion::getTensorViscTerms uses a = 0, b = -1 (Source/Diffusion.cpp:1697-1698).
 * eta_*: face viscosity (1 comp); vel: 3 comps, 1 ghost.
 * lobc/hibc: nbc*3 LinOpBC codes, [n*3+d]; nbc = 1 (all components alike) or 3 (one set per velocity component, the
 * per-component arrays Diffusion::setDomainBC hands to MLTensorOp, Source/Diffusion.cpp:724-731, 1939-2020) */
int iamrx_tensor_apply(const iamrx_geom* g, iamrx_mf out, iamrx_mf vel, double a, double b, iamrx_mf acoef, iamrx_mf eta_x,
                       iamrx_mf eta_y, iamrx_mf eta_z, const int* lobc, const int* hibc, int nbc, int maxorder);
/* implicit Crank-Nicolson solve of Diffusion::diffuse_tensor_velocity (Source/Diffusion.cpp:837-929) */
int iamrx_tensor_solve(const iamrx_geom* g, iamrx_mf soln, iamrx_mf rhs, double a, double b, iamrx_mf acoef, iamrx_mf eta_x,
                       iamrx_mf eta_y, iamrx_mf eta_z, const int* lobc, const int* hibc, int nbc, double tol_rel, double tol_abs,
                       const iamrx_mg_opts* o, iamrx_mg_stats* st);

/* Diffusion::computeExtensiveFluxes on the tensor operator (Source/Diffusion.cpp:1463-1537: MLMG::getFluxes -- the face fluxes of the operator
 * without its b scalar -- times fac x face area), the input of viscflux_reg->FineAdd / CrseInit (:946-954):
 * flux_d(n) = or += fac * area_d * ( -eta_d (4/3 if n == d) du_n/dx_d + cross terms ).  vel: 3 comps, >= 1 ghost cell, as the operator
 * left it after iamrx_tensor_apply(_cf) / iamrx_tensor_solve(_cf) (ghost cells filled with the operator's boundary values). */
int iamrx_tensor_extensive_flux(const iamrx_geom* g, iamrx_mf vel, iamrx_mf eta_x, iamrx_mf eta_y, iamrx_mf eta_z,
                                iamrx_mf flux_x, iamrx_mf flux_y, iamrx_mf flux_z, double fac, int add);
/* the same on a refined level that does not cover the domain (tensorop.setCoarseFineBC(&crsedata, ratio), Source/Diffusion.cpp:733-744,
 * 876-887, 1725-1736; crse_vel == NULL: homogeneous coarse/fine data as in diffuse_tensor_Vsync, :1096-1099).  crse_vel: the coarse
 * level's velocity (3 comps, valid data on its own layout) at the time of the operator. */
int iamrx_tensor_apply_cf(const iamrx_geom* g, iamrx_mf out, iamrx_mf vel, double a, double b, iamrx_mf acoef, iamrx_mf eta_x,
                          iamrx_mf eta_y, iamrx_mf eta_z, const int* lobc, const int* hibc, int nbc, int maxorder,
                          iamrx_mf crse_vel, const iamrx_geom* cgeom, int ratio);
int iamrx_tensor_solve_cf(const iamrx_geom* g, iamrx_mf soln, iamrx_mf rhs, double a, double b, iamrx_mf acoef, iamrx_mf eta_x,
                          iamrx_mf eta_y, iamrx_mf eta_z, const int* lobc, const int* hibc, int nbc, iamrx_mf crse_vel,
                          const iamrx_geom* cgeom, int ratio, double tol_rel, double tol_abs, const iamrx_mg_opts* o, iamrx_mg_stats* st);

/* ---- inter-level data motion (first building blocks) ------------------------------------------------------ */
/* amrex::MultiFab::ParallelCopy: dst(valid + dst_ng) <- src(valid + src_ng) wherever they intersect; dst and src may live on
 * different layouts of the same index space (same index type); periodic_geom != NULL adds the periodic images of src */
int iamrx_parallel_copy(iamrx_mf dst, iamrx_mf src, int scomp, int dcomp, int ncomp, int src_ng, int dst_ng, const iamrx_geom* periodic_geom);
/* amrex::average_down / average_down_faces / average_down_nodal (by index type) as called from NavierStokesBase::avgDown_StatePress
 * (Source/NavierStokesBase.cpp:4125-4193): crse(scomp..) <- mean / injection of fine(scomp..); ratio 2 or 4 */
int iamrx_average_down(iamrx_mf fine, iamrx_mf crse, int scomp, int ncomp, int ratio);
/* NavierStokesBase::create_umac_grown on a refined level (Source/NavierStokesBase.cpp:1108-1311): ghost faces of the fine mac velocities
 * (1 ghost layer) from the coarse ones (FaceLinear) or from neighbouring fine boxes, then IAMR's divergence fix of the outer face of every
 * not-covered ghost cell that has exactly one face neighbour inside the fine level; divu: fine cells with >= 1 ghost, or NULL */
int iamrx_create_umac_grown(iamrx_mf umac_fine_x, iamrx_mf umac_fine_y, iamrx_mf umac_fine_z, iamrx_mf umac_crse_x, iamrx_mf umac_crse_y,
                            iamrx_mf umac_crse_z, iamrx_mf divu, const iamrx_geom* cgeom, const iamrx_geom* fgeom, int ratio);
/* amrex::FluxRegister / YAFluxRegister role for one coarse-fine interface (advective registers: Source/NavierStokesBase.cpp:5036-5096,
 * viscous: Source/NavierStokes.cpp:975-992, Source/Diffusion.cpp:940-953; consumer NavierStokes::reflux, Source/NavierStokes.cpp:1736-1838).
 * Fluxes are extensive (area-weighted) as in IAMR.  crse_init: reg = (add: +=) mult*coarse flux; fine_add: reg += mult * sum of the fine
 * fluxes of a coarse face; reflux: S(outside coarse cell) -=/+= scale*reg/volume on the low/high side of a fine box (periodic images incl.) */
int iamrx_fluxreg_create(iamrx_layout fine, iamrx_layout crse, const iamrx_geom* cgeom, int ratio, int ncomp, iamrx_fluxreg* out);
int iamrx_fluxreg_destroy(iamrx_fluxreg fr);
int iamrx_fluxreg_setval(iamrx_fluxreg fr, double v);
int iamrx_fluxreg_crse_init(iamrx_fluxreg fr, iamrx_mf crse_flux, int dir, int scomp, int dcomp, int ncomp, double mult, int add);
int iamrx_fluxreg_fine_add(iamrx_fluxreg fr, iamrx_mf fine_flux, int dir, int scomp, int dcomp, int ncomp, double mult);
int iamrx_fluxreg_reflux(iamrx_fluxreg fr, iamrx_mf S_crse, double volume, double scale, int scomp, int dcomp, int ncomp);
/* AmrLevel::FillPatch of cell-centred StateData on a refined level = amrex::FillPatchTwoLevels with cell_cons_interp
 * (CellConservativeLinear, linear limiting), the interpolater IAMR registers for State_Type / Gradp_Type (Source/NS_setup.cpp:206-394;
 * call sites e.g. Source/NavierStokesBase.cpp:3382-3418, 4399-4435): dst (fine layout, dst ghost width) <- fine data where a fine box
 * (or its periodic image) exists, conservative-linear interpolation of the coarse data elsewhere inside the domain, physical BC
 * outside; both levels are linearly interpolated in time between their old and new data (old may be NULL).
 * bcrec: ncomp x {lo[3], hi[3]} amrex::BCType codes; extdir_*: [n*3+d] ext_dir values (NS_bcfill.H) or NULL */
int iamrx_fillpatch_two_levels(iamrx_mf dst, int dcomp, double time, iamrx_mf fine_old, iamrx_mf fine_new, double t_fine_old, double t_fine_new,
                               iamrx_mf crse_old, iamrx_mf crse_new, double t_crse_old, double t_crse_new, int scomp, int ncomp,
                               const iamrx_geom* cgeom, const iamrx_geom* fgeom, int ratio, const int* bcrec, const double* extdir_lo, const double* extdir_hi);

/* ---- error estimation and grid generation ---------------------------------------------- */
/* "mag_vort" derived quantity (dermgvort, Source/NS_derive.cpp:86-264, non-EB): out(ocomp) = |curl u|, vel needs 1 filled ghost cell */
int iamrx_derive_mag_vort(const iamrx_geom* g, iamrx_mf out, int ocomp, iamrx_mf vel, int vcomp);
/* one amr.refinement_indicators entry (NavierStokes::error_setup / errorEst, Source/NS_error.cpp:10-145; AMRErrorTag):
 * mode 0 value_greater, 1 value_less, 2 vorticity_greater (value x 2^level), 3 adjacent_difference_greater (field with 1 ghost cell);
 * realbox_lo/hi: in_box_lo / in_box_hi or NULL.  tags: cell-centred, 1 comp; tagged cells are set to 1 */
int iamrx_error_tag(const iamrx_geom* g, iamrx_mf tags, iamrx_mf field, int comp, int mode, double value, int level,
                    const double* realbox_lo, const double* realbox_hi);
/* grid generation from the tags of a level (AmrMesh::MakeNewGrids role: buffer by n_error_buf, Berger-Rigoutsos clustering on the
 * blocking-factor lattice until grid_eff is met, chop to max_grid_size).  Boxes are returned in the index space of the tags
 * (6 ints each: lo, hi); *nboxes in: capacity of boxes, out: number of boxes (error if the capacity is too small). */
int iamrx_cluster_tags(const iamrx_geom* g, iamrx_mf tags, int blocking_factor, int max_grid_size, double grid_eff, int n_error_buf,
                       int* boxes, int* nboxes);
/* The host side of the same on HOST arrays (no device is touched): tags / allowed: one byte per cell of [dom_lo, dom_hi], x fastest; allowed
 * (may be NULL): cells the new level may cover (Amr::regrid(lbase > 0): the proper nesting domain) -- a block of the blocking-factor
 * lattice counts only if all of its cells are allowed.  iamrx_host_erode: the erosion that forms that domain (a cell survives `passes`
 * passes if its 26 neighbours do; periodic images count, nothing constrains beyond a non-periodic face), in place. */
int iamrx_host_cluster_tags(const unsigned char* tags, const int dom_lo[3], const int dom_hi[3], int blocking_factor, int max_grid_size, double grid_eff,
                            int n_error_buf, const unsigned char* allowed, int* boxes, int* nboxes);
int iamrx_host_erode(unsigned char* map, const int n[3], const int periodic[3], int passes);

/* ---- Diffusion operator entries on caller-owned data (the operator-level boundary) -------------------------------------
 * What NavierStokes::scalar_diffusion_update / velocity_diffusion_update / mac_sync hand to the reference's Diffusion class, for a host
 * code that keeps its own state.  All arrays are the caller's; the level's own time step (iamrx_ns_advance) goes through the same code.
 * LinOp BC types (lobc / hibc): as iamrx_abec_* (Diffusion::setDomainBC, Source/Diffusion.cpp:1886-1941); per velocity component for
 * the tensor entries ([n*3+d]).  crse: the coarse level of a refined level, NULL on level 0: its state at the old / new time (valid cells
 * on its own layout, same component numbering as the fine arrays), crse_new NULL = homogeneous coarse/fine data. */
typedef struct iamrx_diffusion_crse { iamrx_mf crse_old, crse_new; const iamrx_geom* cgeom; int ratio; } iamrx_diffusion_crse;
/* Diffusion::diffuse_scalar (Source/Diffusion.cpp:207-599; declaration Source/Diffusion.H:79-104): Crank-Nicolson update of component sigma,
 *   (alpha - theta dt div beta grad) s_new = alpha s* + (1 - theta) dt div beta grad s_old + dt delta_rhs,
 * rho_flag 0: s = S, alpha = 1; 1: alpha = rho_half; 2: s = S / rho, alpha = rho_new, S_new = s rho_new.  S_old / S_new: >= 1 ghost cell
 * FILLED by the caller (FillPatch, Source/Diffusion.cpp:237-239); Rho_old / Rho_new (NULL: S_old / S_new) hold the density in rho_comp;
 * fluxn / fluxnp1 (3 face arrays each, or NULL): the extensive fluxes (1 - theta) area (-beta grad s_old), theta area (-beta grad s_new);
 * delta_rhs (NULL: none); betan / betanp1: 3 face arrays of the diffusivity (betan NULL with theta = 1 or add_old_time_divFlux = 0). */
int iamrx_diffuse_scalar(const iamrx_geom* g, iamrx_mf S_old, iamrx_mf Rho_old, iamrx_mf S_new, iamrx_mf Rho_new, int sigma, int rho_comp, double dt,
                         double be_cn_theta, iamrx_mf rho_half, int rho_flag, const iamrx_mf* fluxn, const iamrx_mf* fluxnp1, iamrx_mf delta_rhs,
                         int rhs_comp, const iamrx_mf* betan, const iamrx_mf* betanp1, const int* lobc, const int* hibc,
                         const iamrx_diffusion_crse* crse, int add_old_time_divFlux, double visc_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);
/* Diffusion::diffuse_tensor_velocity (Source/Diffusion.cpp:617-957; Source/Diffusion.H:117-127): (alpha - theta dt div tau) u_new =
 * alpha u* + (1 - theta) dt div tau(u_old), alpha = rho_half (rho_flag 1) or rho_new with u* weighted by rho_old (rho_flag 3).  U_old / U_new:
 * states with the velocity in components 0..2 and the density in rho_comp, 1 filled ghost cell.  visc_old_term (NULL: evaluate): div tau(u_old).
 * tflux (3 face arrays of 3 components, or NULL).  fill_new (NULL: none): called once U_new's velocity holds rho u*, to refill its ghost
 * cells (the FillPatch of Source/Diffusion.cpp:866). */
int iamrx_diffuse_tensor_velocity(const iamrx_geom* g, iamrx_mf U_old, iamrx_mf U_new, int rho_comp, double dt, double be_cn_theta, iamrx_mf rho_half,
                                  int rho_flag, iamrx_mf visc_old_term, const iamrx_mf* eta_n, const iamrx_mf* eta_np1, const int* lobc, const int* hibc,
                                  const iamrx_diffusion_crse* crse, const iamrx_mf* tflux, double visc_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st,
                                  void (*fill_new)(void* ctx, iamrx_mf U_new), void* ctx);
/* Diffusion::diffuse_Vsync -> diffuse_tensor_Vsync (Source/Diffusion.cpp:960-1178): Vsync (3 components, 1 ghost cell) is replaced by the
 * solution of (alpha - theta dt div tau) V = rho Vsync with homogeneous boundary and coarse/fine data (cgeom NULL: level 0); eta: the face
 * coefficients (upstream passes ones, :1122-1135); bcrec_vel: the BCRecs of the velocity (6 ints per component), ghost cells outside
 * ext_dir faces end up zero (:987-1008); tflux: theta area (-tau), or NULL. */
int iamrx_diffuse_tensor_vsync(const iamrx_geom* g, iamrx_mf Vsync, double dt, double be_cn_theta, iamrx_mf rho_half, int rho_flag, iamrx_mf Rho_old,
                               iamrx_mf Rho_new, int rho_comp, const iamrx_mf* eta, const int* lobc, const int* hibc, const int* bcrec_vel,
                               const iamrx_geom* cgeom, int ratio, const iamrx_mf* tflux, double visc_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);
/* Diffusion::diffuse_Ssync as NavierStokes::mac_sync calls it (Source/NavierStokes.cpp:1590-1640, Source/Diffusion.cpp:1181-1352): component
 * comp of Ssync (a rate) becomes the diffused sync increment, (alpha - theta dt div beta grad) s = dt Ssync (x rho_half for rho_flag 1),
 * Ssync = s (x rho_new for rho_flag 2); flux: theta area (-beta grad s), or NULL. */
int iamrx_diffuse_ssync(const iamrx_geom* g, iamrx_mf Ssync, int comp, double dt, double be_cn_theta, iamrx_mf rho_half, int rho_flag, iamrx_mf Rho_new,
                        int rho_comp, const iamrx_mf* beta, const int* lobc, const int* hibc, const iamrx_geom* cgeom, int ratio, const iamrx_mf* flux,
                        double visc_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);

/* ---- level time step (NavierStokes::advance and the init sequence) ----------------------------------- */
typedef struct iamrx_ns_params {
    double cfl, visc_coef, be_cn_theta, gravity;
    double mac_tol, mac_abs_tol, proj_tol, proj_abs_tol, visc_tol;
    int use_forces_in_trans, do_mom_diff, init_iter, init_vel_iter;
    double init_shrink, change_max, fixed_dt;
    int nscal, verbose;
    double init_dt;              /* ns.init_dt: dt when the state has no velocity/force scale (LidDrivenCavity start) */
    double tracer_diff_coef;     /* ns.scal_diff_coefs[0] (<= 0: tracer not diffusive) */
    int phys_lo[3], phys_hi[3];  /* ns.lo_bc / ns.hi_bc, Source/NS_BC.H: 0 Interior (periodic), 1 Inflow, 2 Outflow, 3 Symmetry, 4 SlipWall, 5 NoSlipWall */
    double wall_vel_lo[9], wall_vel_hi[9];   /* xlo.velocity ... zhi.velocity: [d*3+n] = component n on the lo/hi face of direction d */
    double scal_bc_lo[12], scal_bc_hi[12];   /* xlo.density, xlo.tracer, xlo.tracer2, xlo.temp ... zhi.* (inflow values): [d*4+n], n = the scalar's slot: 0 density, 1 tracer, then tracer2 / temp as present */
    int do_cons_trac;            /* ns.do_cons_trac: the tracer is rho*q, advected conservatively and diffused as div beta grad(S/rho) (Source/NS_setup.cpp:306-310) */
    int do_denminmax, do_scalminmax;   /* ns.do_denminmax / ns.do_scalminmax (Source/NavierStokesBase.cpp:466-467): clip the advected density / scalars to the 27-point min / max of the old data (ConservativeScalMinMax / ConvectiveScalMinMax, :4256-4368) */
    int do_trac2, do_cons_trac2; /* ns.do_trac2 / ns.do_cons_trac2: a second tracer, state component 5 (Source/NavierStokes.cpp:45-46, Source/NS_setup.cpp:312-320) */
    double tracer2_diff_coef;    /* ns.scal_diff_coefs[1] */
    int do_temp;                 /* ns.do_temp: temperature as the last state component (Source/NavierStokes.cpp:47-48; temp_bc, RhoInverse_Laplacian_S) */
    double temp_cond_coef;       /* ns.temp_cond_coef */
    int use_ppm;                 /* ns.advection_scheme: 0 = Godunov_PLM, 1 = Godunov_PPM, 2 = BDS (Source/NavierStokesBase.cpp:548-553, 4654-4656) */
} iamrx_ns_params;
void iamrx_ns_default_params(iamrx_ns_params* p);     /* defaults of Source/NavierStokesBase.cpp:96-170 */
int iamrx_ns_create(const iamrx_geom* g, iamrx_layout l, const iamrx_ns_params* p, const iamrx_mg_opts* o, iamrx_ns* out);
int iamrx_ns_destroy(iamrx_ns ns);
int iamrx_ns_init_taylorgreen(iamrx_ns ns, double vfac, double a, double b, double c, double rho0);  /* Source/prob/prob_init.cpp:509-560 */
/* prob.probtype = 10, RayleighTaylor (Source/prob/prob_init.cpp:407-488) */
int iamrx_ns_init_rayleightaylor(iamrx_ns ns, double rho_1, double rho_2, double tra_1, double tra_2, double pertamp, double interface_width);
int iamrx_ns_init_rest(iamrx_ns ns, double rho0);          /* probtype 1, LidDrivenCavity (Source/prob/prob_init.cpp:102-109) */
int iamrx_ns_post_init(iamrx_ns ns, double stop_time);     /* NavierStokes::post_init (Source/NavierStokes.cpp:1254-1299) */
int iamrx_ns_step(iamrx_ns ns, double* dt_used);           /* computeNewDt + NavierStokes::advance (Source/NavierStokes.cpp:543-691) */
int iamrx_ns_advance(iamrx_ns ns, double dt, double* dt_est);
int iamrx_ns_time(iamrx_ns ns, double* time, double* dt, int* nstep);
/* snapshot COPY (caller destroys it with iamrx_mf_destroy) of a persistent array: 0 S_new, 1 S_old, 2 P_new,
 * 3 P_old, 4 Gp_new, 5 Gp_old, 6..8 u_mac, 9 aofs  (get_new_data/get_old_data role); 10, 11: the last two MAC potentials (the
 * initial-guess history of the MAC solve, part of a checkpoint) */
int iamrx_ns_data(iamrx_ns ns, int which, iamrx_mf* out);
/* Derived quantities of the plotfile (derive_lst, Source/NS_setup.cpp:436-449; amr.derive_plot_vars): "energy" = rho |u|^2 / 2 (derkeng,
 * Source/NS_derive.cpp:266-295), "mag_vort" = |curl u| (dermgvort, :86-264, ghost cells by FillPatch), "avg_pressure" = mean of the eight
 * nodes of the cell (deravgpres, :51-80) of the level's new-time data, into component ocomp of the cell-centred out (the level's layout). */
int iamrx_ns_derive(iamrx_ns ns, const char* name, iamrx_mf out, int ocomp);

/* overwrite state (0,1), pressure (2,3) or grad p (4,5) with src (same layout and ngrow; ncomp at most the array's: the leading
 * components are set): the role of NavierStokes::initData for caller-supplied initial data (Source/NavierStokes.cpp:318-420).
 * The state arrays hold u v w density tracer [tracer2] [temp], and with ns.do_temp two more components, divu and dsdt (the
 * reference's Divu_Type / Dsdt_Type), which the library computes. */
int iamrx_ns_set_data(iamrx_ns ns, int which, iamrx_mf src);      /* which: 0..5, 10, 11 */
/* checkpoint / restart of one level (AmrLevel::checkPoint / NavierStokesBase::restart role, Source/NavierStokesBase.cpp:856-897,
 * 2706-2727): what outlives a time step besides the arrays of iamrx_ns_data.  state[16] = time, dt, nstep, State_Type new / old time,
 * Press_Type new interval [2], old interval [2], dt of the last MAC solve, two history flags, dt estimate of the last advance,
 * stop_time, 2 unused.  set = 0: read; set = 1: write (after the arrays have been set with iamrx_ns_set_data; the level then continues
 * as if it had taken the steps itself -- no post_init). */
int iamrx_ns_restart_state(iamrx_ns ns, int set, double state[16]);
int iamrx_ns_stats(iamrx_ns ns, iamrx_mg_stats* mac, iamrx_mg_stats* nodal, iamrx_mg_stats* visc);
/* per-section wall time accumulation (ms): predict, mac, advect, update, viscous, nodal; enable=1 inserts stream syncs (2: and resets).
 * enable=3: no syncs; HIP events on the launch stream around every 8th k_nodal_gs4 launch of the level's own (finest) MG level until the
 * next call, which returns their total duration (ms) in sections_ms[6] and their number in sections_ms[7] */
int iamrx_ns_profile(iamrx_ns ns, int enable, double sections_ms[8]);

/* ---- zero-copy view of caller-owned device memory ----------------------------------------------------------------------
 * AMReX keeps its FABs in device memory on GPU builds (The_Arena); with the same Array4 layout on both sides a MultiFab needs no
 * copy at the seam: dev_ptrs[li] = FArrayBox::dataPtr() of the li-th LOCAL box (allocated region = valid box converted to `type`
 * and grown by ngrow).  The library never frees, moves or reallocates the memory; destroy the handle with iamrx_mf_destroy. */
int iamrx_mf_alias(iamrx_layout l, const int type[3], int ncomp, int ngrow, double* const* dev_ptrs, iamrx_mf* out);

/* Projection::level_project on a level that covers the domain (Source/Projection.cpp:166-450; declaration Source/Projection.H:53-75), the
 * call NavierStokesBase::level_projector makes (Source/NavierStokesBase.cpp:1894-1924): P_new = 0; U_new /= dt; U_new += Gp_old/rho_half;
 * sigma = 1/rho_half; doMLMGNodalProjection (Gp_new = grad phi, P_new = phi); U_new *= dt.  In place, like the reference.
 * U_new: velocity at comps vcomp..vcomp+2, 1 ghost cell (inflow data in the ghost cells of inflow faces, in U/dt units);
 * lobc/hibc: LinOpBC codes of Projection.cpp:2432-2464 (102 Neumann, 101 outflow, 103 inflow). */
int iamrx_level_project(const iamrx_geom* g, double dt, iamrx_mf U_new, int vcomp, iamrx_mf P_new, iamrx_mf Gp_old, iamrx_mf Gp_new,
                        iamrx_mf rho_half, const int lobc[3], const int hibc[3], double proj_tol, double proj_abs_tol,
                        const iamrx_mg_opts* o, iamrx_mg_stats* st);

/* SyncRegister (Source/SyncRegister.H:40-49, Source/SyncRegister.cpp): nodal register on the faces of the coarsened fine boxes.
 * crse_init: SyncRegister::CrseInit(Sync_resid_crse, crse_geom, mult) (:306-318); fine_add: FineAdd(Sync_resid_fine, crse_geom, mult)
 * (:350-607; the fine residual needs no ghost nodes here); init_rhs: InitRHS(rhs, geom, phys_bc) (:47-304; rhs: nodal, coarse layout).
 * phys_lo/hi: PhysBCType codes of ns.lo_bc / ns.hi_bc (outflow faces are zeroed in InitRHS). */
int iamrx_syncreg_create(iamrx_layout fine, iamrx_layout crse, const iamrx_geom* cgeom, const iamrx_geom* fgeom, int ratio,
                         const int phys_lo[3], const int phys_hi[3], iamrx_syncreg* out);
int iamrx_syncreg_destroy(iamrx_syncreg r);
int iamrx_syncreg_crse_init(iamrx_syncreg r, iamrx_mf sync_resid_crse, double mult);
int iamrx_syncreg_fine_add(iamrx_syncreg r, iamrx_mf sync_resid_fine, double mult);
int iamrx_syncreg_init_rhs(iamrx_syncreg r, iamrx_mf rhs);
/* Projection::MLsyncProject (Source/Projection.cpp:457-607; declaration Source/Projection.H:99-118) on caller-owned data: the two-level sync
 * projection NavierStokesBase::level_sync drives (Source/NavierStokesBase.cpp:1927-2044).  A level of the solve: its geometry and boxes,
 * the nodal LinOp BC (lobc / hibc: periodic / Neumann / Dirichlet at outflow / inflow, Source/Projection.cpp:2432-2464), the ratio to the
 * next coarser level, gp (or NULL): the Gradp array that accumulates grad(phi).
 * Vsync (coarse, 3 components, 1 ghost cell) and V_corr (fine: Vsync interpolated with iamrx_sync_interp) are projected with
 * sigma = 1 / rho_crse, 1 / rho_fine and the right-hand side of rhs_sync_reg (SyncRegister::InitRHS); phi_crse / phi_fine (nodal, 1 ghost
 * cell) return the pressure correction, which is also added to pres_crse / pres_fine; vel_* (velocity at vcomp_*) += dt * the projected
 * increments.  crse_sync_reg (NULL on level 0): the register of the interface BELOW the coarse level, which receives the residual of the
 * composite solution on that level's boundary when crse_iteration == crse_dt_ratio (SyncRegister::CompAdd, Source/SyncRegister.cpp:302-348). */
typedef struct iamrx_proj_level { const iamrx_geom* geom; iamrx_layout layout; int lobc[3], hibc[3]; int ratio; iamrx_mf gp; } iamrx_proj_level;
int iamrx_mlsync_project(const iamrx_proj_level* crse, const iamrx_proj_level* fine, iamrx_mf pres_crse, iamrx_mf vel_crse, int vcomp_crse, iamrx_mf pres_fine,
                         iamrx_mf vel_fine, int vcomp_fine, iamrx_mf rho_crse, iamrx_mf rho_fine, iamrx_mf Vsync, iamrx_mf V_corr, iamrx_mf phi_crse,
                         iamrx_mf phi_fine, iamrx_syncreg rhs_sync_reg, iamrx_syncreg crse_sync_reg, double dt, int crse_iteration, int crse_dt_ratio,
                         double sync_tol, double abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);
/* NavierStokesBase::SyncInterp with cell_cons_interp (Source/NavierStokesBase.cpp:3071-3276): fine_dst(dcomp..) = conservative-linear
 * interpolant of crse_sync(scomp..) on every cell of the fine level; the caller applies `increment` / dt_clev (:3215-3262) */
int iamrx_sync_interp(iamrx_mf fine_dst, int dcomp, iamrx_mf crse_sync, int scomp, int ncomp, const iamrx_geom* cgeom, const iamrx_geom* fgeom,
                      int ratio, const int* bcrec /* [ncomp][6] */);
/* NavierStokesBase::ComputeAofs with is_sync = true as MacProj::mac_sync_compute calls it (Source/MacProj.cpp:700-731,
 * Source/NavierStokesBase.cpp:4681-4683, 4777, 4826-4832): edge states traced with u_mac, fluxes = edge * Ucorr * area,
 * sync(acomp..) -= -div(F)/vol; flux_* (optional) receive the fluxes for the flux registers */
int iamrx_godunov_compute_aofs_sync(const iamrx_geom* g, iamrx_mf sync, int acomp, iamrx_mf S, int ncomp, iamrx_mf force, iamrx_mf divu,
                                    iamrx_mf umac_x, iamrx_mf umac_y, iamrx_mf umac_z, iamrx_mf ucorr_x, iamrx_mf ucorr_y, iamrx_mf ucorr_z,
                                    const int* iconserv, double dt, const int* bcrec, int is_velocity, int use_forces_in_trans,
                                    iamrx_mf flux_x, iamrx_mf flux_y, iamrx_mf flux_z, int scheme);

/* SyncRegister::CompAdd (Source/SyncRegister.H:45, SyncRegister.cpp:302-348): sync_resid_fine (nodal, on the fine side of r's interface; the
 * residual a sync projection of the levels above leaves on that level) is zeroed on the nodes of the boxes of `finer` (the next finer level,
 * ratio finer_ratio to the residual's level whose geometry is fgeom; periodic images included: upstream's Pgrids) and added like FineAdd
 * with `mult`.  The residual array is modified, as upstream's is. */
int iamrx_syncreg_comp_add(iamrx_syncreg r, iamrx_mf sync_resid_fine, const iamrx_geom* fgeom, iamrx_layout finer, int finer_ratio, double mult);
/* MacProj::mac_sync_compute, the form NavierStokes::mac_sync calls (Source/MacProj.H:60-75, MacProj.cpp:488-731), on caller-owned arrays of one
 * level: the velocity forcing gravity * rho (z) + visc_vel - gradp, divided by rho unless do_mom_diff (:598-640), then ComputeAofs with
 * is_sync = true for the three velocities (Vsync -= update) and for the nscal scalars starting with the density (Ssync -= update), both
 * traced with umac and fluxed with ucorr.  S_vel (3 comps; rho u under do_mom_diff, :536-553) / S_scal (nscal comps): the state at
 * prev_time with 3 filled ghost cells; visc_vel (3) / tforce_scal (nscal, the scalars' forcing as :641-683 assemble it) / gradp (3) / divu
 * (1): 1 ghost cell, visc_vel / tforce_scal / divu may be NULL (zero).  fluxv_* (3 comps) / fluxs_* (nscal comps), optional: the fluxes
 * for the caller's advective registers (CrseInit / FineAdd with the sync sign, Source/NavierStokesBase.cpp:5083-5096; the FineAdd of ucorr
 * to the mac register, MacProj.cpp:707-727, is the caller's as well: iamrx_fluxreg_fineadd). */
int iamrx_mac_sync_compute(const iamrx_geom* g, iamrx_mf ucorr_x, iamrx_mf ucorr_y, iamrx_mf ucorr_z, iamrx_mf Vsync, iamrx_mf Ssync, iamrx_mf S_vel,
                           iamrx_mf S_scal, int nscal, iamrx_mf visc_vel, iamrx_mf tforce_scal, iamrx_mf gradp, iamrx_mf divu, iamrx_mf umac_x,
                           iamrx_mf umac_y, iamrx_mf umac_z, const int* iconserv_scal, int do_mom_diff, double gravity, double dt, const int* bcrec_vel,
                           const int* bcrec_scal, int use_forces_in_trans, int scheme, iamrx_mf fluxv_x, iamrx_mf fluxv_y, iamrx_mf fluxv_z,
                           iamrx_mf fluxs_x, iamrx_mf fluxs_y, iamrx_mf fluxs_z);
/* MacProj::mac_sync_compute, the form with the half-time edge states handed in (Source/MacProj.H:77-94, MacProj.cpp:733-786; one component):
 * flux = edge(edge_comp) * ucorr * area, Sync(sync_indx) -= -div(flux) / vol; flux_* (optional, 1 comp) return the fluxes */
int iamrx_mac_sync_compute_edge(const iamrx_geom* g, iamrx_mf ucorr_x, iamrx_mf ucorr_y, iamrx_mf ucorr_z, iamrx_mf Sync, int sync_indx, iamrx_mf edge_x,
                                iamrx_mf edge_y, iamrx_mf edge_z, int edge_comp, iamrx_mf flux_x, iamrx_mf flux_y, iamrx_mf flux_z);
/* Projection::initialVelocityProject (Source/Projection.H:116-121, Projection.cpp:615-838) on caller-owned arrays of the levels
 * levels[0 .. nlev-1] (coarsest first): pres[l] (nodal, 1 ghost) is zeroed and returns phi; vel[l] (velocity at vcomp[l], 1 ghost cell with
 * the inflow data) is projected on the composite grid, div(sigma grad phi) = div(vel) - <divu>, vel -= sigma grad phi, sigma = 1
 * (rho == NULL: rho_wgt_vel_proj = 0) or 1 / rho[l](rho_comp[l]); divu / divu_comp: the constraint of a variable-divergence run or NULL;
 * levels[l].gp (optional) receives grad phi. */
int iamrx_initial_velocity_project(int nlev, const iamrx_proj_level* levels, const iamrx_mf* vel, const int* vcomp, const iamrx_mf* pres, const iamrx_mf* rho,
                                   const int* rho_comp, const iamrx_mf* divu, const int* divu_comp, double proj_tol, double proj_abs_tol,
                                   const iamrx_mg_opts* o, iamrx_mg_stats* st);
/* Projection::initialSyncProject (Source/Projection.H:123-134, Projection.cpp:970-1185) on caller-owned arrays: vel_new <- (vel_new - vel_old) / dt,
 * sigma = 1 / rho_half, velocities averaged down, rhcc = -(divu_new - divu_old) / dt (NULL: none), composite projection that ACCUMULATES
 * grad phi in levels[l].gp; phi[l] (the caller's old-time pressure array, zeroed first) returns the correction, which is also added to
 * pres_new[l] (may be NULL).  vel_new keeps the projected acceleration (upstream resets the state afterwards, NavierStokesBase::resetState). */
int iamrx_initial_sync_project(int nlev, const iamrx_proj_level* levels, const iamrx_mf* vel_new, const int* vcomp, const iamrx_mf* vel_old, const iamrx_mf* phi,
                               const iamrx_mf* pres_new, const iamrx_mf* rho_half, const iamrx_mf* divu_new, const iamrx_mf* divu_old, const int* divu_comp,
                               double dt, double proj_tol, double proj_abs_tol, const iamrx_mg_opts* o, iamrx_mg_stats* st);

/* ---- multi-level time step ------------------------------------------------------------------------------------
 * One coarse time step of a hierarchy of levels with subcycling = amrex::Amr::coarseTimeStep -> timeStep(level): advance(level),
 * ncycle x timeStep(level+1), NavierStokesBase::post_timestep(level) (Source/NavierStokesBase.cpp:2546-2636):
 *   NavierStokes::reflux (Source/NavierStokes.cpp:1736-1838), avgDown (:1845-1873), mac_sync (:1438-1730) with
 *   MacProj::mac_sync_solve / mac_sync_compute (Source/MacProj.cpp:359-731), NavierStokesBase::level_sync (Source/NavierStokesBase.cpp:1927-2044)
 *   with Projection::MLsyncProject (Source/Projection.cpp:457-607), SyncRegister (Source/SyncRegister.cpp), SyncInterp (:3071-3276);
 * and NavierStokes::post_init for all levels (initialVelocityProject, initialPressureProject, init_iter x {advance, initialSyncProject},
 * Source/NavierStokes.cpp:1254-1432, Source/Projection.cpp:615-1185).
 * layouts[l]: the boxes of level l in the index space of that level (level 0 must cover g0's domain; finer levels must be aligned to
 * the refinement ratio and properly nested).  ratio: 2.  The level handles returned by iamrx_amr_level are borrowed: use them with
 * iamrx_ns_data / iamrx_ns_set_data / iamrx_ns_time / iamrx_ns_stats, do not destroy them. */
int iamrx_amr_create(const iamrx_geom* g0, int nlev, const iamrx_layout* layouts, int ratio, const iamrx_ns_params* p, const iamrx_mg_opts* o, iamrx_amr* out);
int iamrx_amr_destroy(iamrx_amr a);
/* ---- regridding: Amr::regrid from level 0 with IAMR's error estimation (amr.refinement_indicators, Source/NS_error.cpp:10-145) and
 * NavierStokesBase::init(AmrLevel&) / init() for the data of the new levels (Source/NavierStokesBase.cpp:1713-1806).
 * comp: state component 0..4 (velocity, density, tracer) or -1 = mag_vort; mode: 0 value_greater, 1 value_less, 2 vorticity_greater
 * (x 2^level), 3 adjacent_difference_greater; value[nvalue]: per level (the last one repeats); tags only on levels < max_level;
 * has_box: in_box_lo / in_box_hi.  regrid_int > 0: iamrx_amr_coarse_step regrids at the start of every regrid_int-th coarse step.
 * After a regrid (check iamrx_amr_nlevels / iamrx_amr_level_boxes, or simply re-fetch after every coarse step) the level handles of
 * iamrx_amr_level must be fetched again: handles given out before stay allocated until iamrx_amr_destroy but are retired -- every
 * iamrx_ns_* call on them returns an error ("stale level handle") instead of touching a freed level. */
typedef struct iamrx_tag_rule { int comp, mode, nvalue, max_level, has_box; double value[8]; double box_lo[3], box_hi[3]; } iamrx_tag_rule;
int iamrx_amr_set_regrid(iamrx_amr a, int max_level, int regrid_int, int blocking_factor, int max_grid_size, double grid_eff, int n_error_buf,
                         int nrules, const iamrx_tag_rule* rules);
/* The regrids of the last iamrx_amr_coarse_step.  As in Amr::timeStep, at the start of every step of every level each level i from there up
 * whose own step count since the grids above it were last rebuilt has reached regrid_int rebuilds the levels above it (regrid(i)), so a
 * regrid can start above level 0 and fall inside a coarse step.  Event e: the base level (the levels up to it kept their grids), the
 * time, the number of rebuilt levels and, level by level from lbase + 1, their boxes (nboxes[l], then 6 ints per box; boxes NULL: sizes
 * only).  For drivers that mirror the hierarchy elsewhere. */
int iamrx_amr_regrid_log_count(iamrx_amr a, int* nevents);
int iamrx_amr_regrid_log_event(iamrx_amr a, int event, int* lbase, double* time, int* nlevels, int* nboxes /* [8] */, int* boxes);
/* amr.compute_new_dt_on_regrid (default 0): 1 = after a regrid from level 0 that changed the grids the time steps are recomputed with
 * NavierStokesBase::computeNewDt(post_regrid_flag = 1) (Source/NavierStokesBase.cpp:971-982), as Amr::timeStep does; 0 = levels that
 * existed keep their dt, new levels start with dt_level[l-1] / n_cycle[l]. */
int iamrx_amr_set_compute_new_dt_on_regrid(iamrx_amr a, int on);
/* NavierStokesBase::manual_tags_placement (Source/NavierStokesBase.cpp:2112-2215; ns.do_refine_outflow = 0, ns.do_derefine_outflow = 1,
 * ns.Nbuf_outflow = 1 are the defaults here as upstream): with an outflow face, either refine the whole layer next to it once it holds a
 * tag, or keep Nbuf_outflow level-0 cells (rounded up to the blocking factor, grown level by level) next to it unrefined.  Call after
 * iamrx_amr_set_regrid. */
int iamrx_amr_set_outflow_tagging(iamrx_amr a, int do_refine_outflow, int do_derefine_outflow, int nbuf_outflow);
int iamrx_amr_regrid(iamrx_amr a, int* changed);
/* install given grids of levels 1 .. nfine_levels (boxes in each level's own index space, 6 ints per box, level by level) and fill them */
int iamrx_amr_install_grids(iamrx_amr a, int nfine_levels, const int* nboxes, const int* boxes, int* changed);
int iamrx_amr_nlevels(iamrx_amr a, int* nlev);
/* a handle of the layout level `lev` currently lives on (release it with iamrx_layout_destroy): for containers that exchange data with the level */
int iamrx_amr_level_layout(iamrx_amr a, int lev, iamrx_layout* out);
int iamrx_amr_level_boxes(iamrx_amr a, int lev, int* nboxes, int* boxes /* NULL: query the count */);
int iamrx_amr_level(iamrx_amr a, int lev, iamrx_ns* out);
int iamrx_amr_post_init(iamrx_amr a, double stop_time);      /* S_new of every level must hold the initial data (iamrx_ns_set_data) */
/* Amr::checkPoint / Amr::restart role for the hierarchy: dt_level, dt_min, n_cycle per level, counters = {level_steps, level_count},
 * stop_time.  set = 0: read, 1: write (instead of iamrx_amr_post_init, after every level has been restored). */
int iamrx_amr_restart_state(iamrx_amr a, int set, double* dt_level, double* dt_min, int* n_cycle, int counters[2], double* stop_time);
/* Amr::level_count of every level (the steps of level i since the last regrid that rebuilt it; Amr::timeStep regrids from level i when it
 * reaches regrid_int), n = amr.max_level + 1 entries as Amr::checkPoint writes them.  set = 0: read (0 beyond the levels that exist) */
int iamrx_amr_level_counts(iamrx_amr a, int set, int* counts, int n);
/* amr.restart takes stop_time from the inputs file, not from the checkpoint (Amr::restart re-reads it): the level's own copy, which
 * clips its time step (NavierStokesBase::computeNewDt, Source/NavierStokesBase.cpp:1011-1023) */
int iamrx_ns_set_stop_time(iamrx_ns ns, double stop_time);
int iamrx_amr_coarse_step(iamrx_amr a, double* dt0);         /* dt0: the level-0 time step used */
int iamrx_amr_time(iamrx_amr a, double* time, double* dt_levels /* [nlev] or NULL */);
/* the pieces of NavierStokesBase::post_timestep(lev) one by one (lev < finest), for a caller that drives the loop itself:
 * NavierStokes::reflux, avgDown, mac_sync (= MacProj::mac_sync_solve + mac_sync_compute + the state update and SyncInterp),
 * NavierStokesBase::level_sync (= SyncInterp + Projection::MLsyncProject) */
int iamrx_amr_reflux(iamrx_amr a, int lev);
int iamrx_amr_avg_down(iamrx_amr a, int lev);
int iamrx_amr_mac_sync(iamrx_amr a, int lev);
int iamrx_amr_level_sync(iamrx_amr a, int lev);
int iamrx_amr_sync_stats(iamrx_amr a, iamrx_mg_stats* sync_project, iamrx_mg_stats* mac_sync);
/* section profile of the coarse steps (synchronising; measurement aid like iamrx_ns_profile).  Reads the accumulated times first, then
 * enable: 1 = reset and start, 0 = stop, -1 = leave as is.  sections_ms[16]: [0] reflux, [1] avgDown, [2] mac_sync_solve, [3] rest of mac_sync
 * (mac_sync_compute, viscous / scalar sync solves, SyncInterp), [4] level_sync (MLsyncProject), [5] regrid, [8 + l] advance of level l;
 * level_sections_ms[8 * l + i]: the sections of iamrx_ns_profile of level l (may be NULL). */
int iamrx_amr_profile(iamrx_amr a, int enable, double sections_ms[16], double level_sections_ms[]);

#ifdef __cplusplus
}
#endif
#endif
