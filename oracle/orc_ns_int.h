/* oracle/orc_ns_int.h -- the level object shared by orc_ns.c (NavierStokes::advance on one level) and orc_amr.c (the multi-level
 * driver and the sync operations).  Test infrastructure only, see orc.h.
 *
 * Representation of an AMR level > 0: every array of the level is a WHOLE-DOMAIN array of the level's index space; the level
 * itself is the union of `nbox` boxes, `cov` is 1 on its cells.  Entries outside the level hold data that no result depends on
 * (FillPatch'ed copies, see ns_fillpatch).  Registers of a coarse/fine interface are owned by the FINE level, as in IAMR
 * (NavierStokesBase.H:700-706: advflux_reg, viscflux_reg, sync_reg; MacProj::mac_reg[level]), and live on the index space of the
 * COARSE level: one value per coarse face (flux registers) / coarse node (sync register). */
#ifndef ORC_NS_INT_H
#define ORC_NS_INT_H
#include "orc_int.h"

enum { Xvel = 0, Density = 3, Tracer = 4, ORC_MAXSCAL = 4, ORC_MAXSLOT = ORC_MAXSCAL + 2 };   /* Tracer2 / Temp: s->Tracer2 / s->Temp (-1: absent), NavierStokes.cpp:43-48 */

struct orc_ns_state {
    orc_geom g;
    orc_ns_params p;
    orc_mg_opts o;
    orc_fab S[2];      /* [new, old] swapped by index */
    orc_fab P[2];
    orc_fab Gp[2];
    int inew;          /* index of "new" for S */
    int pnew;          /* index of "new" for P/Gp */
    orc_fab umac[3];
    orc_fab aofs;
    orc_fab rho_ptime, rho_ctime, rho_half;
    double time, dt, dt_min_adv;
    int nstep;
    int initial_step, initial_iter;
    orc_mg_stats st_mac, st_nodal, st_visc, st_scal;
    int lobc[3], hibc[3];          /* LinOp BC of the MAC projection: Neumann at walls / inflow, Dirichlet at outflow */
    int nlobc[3], nhibc[3];        /* nodal projection: the same with ORC_LO_INFLOW on inflow faces */
    int nstate, nscal;             /* NUM_STATE, NUM_SCALARS = NUM_STATE - Density (NavierStokes.cpp:43-55) */
    int have_divu;                 /* ns.do_temp: Divu_Type and Dsdt_Type exist (NS_setup.cpp:365-383).  They are Point-type cell data with the
                                    * times of State_Type and are kept here as two more components of the S arrays -- Divu = nstate,
                                    * Dsdt = nstate + 1, nalloc = nstate + 2 -- with their own BCRecs in the slots after the scalars */
    int nalloc, Divu, Dsdt;
    int Tracer2, Temp;             /* state components, -1 when ns.do_trac2 / ns.do_temp are off */
    int scal_cons[ORC_MAXSCAL];    /* advectionType == Conservative (NS_setup.cpp:297-320) per scalar slot (0 = density) */
    int scal_rho_flag[ORC_MAXSCAL];/* Diffusion::set_rho_flag(diffusionType): 0 Laplacian_S, 1 RhoInverse_Laplacian_S (Temp), 2 Laplacian_SoverRho */
    double scal_diff[ORC_MAXSCAL]; /* visc_coef[Density + n] (<= 0: not diffusive) */
    double ed_scal_lo[3 * ORC_MAXSLOT], ed_scal_hi[3 * ORC_MAXSLOT];   /* ext_dir (inflow) values [n*3+d] of density, tracer, ... */
    orc_bcrec bc_vel[3], bc_scal[ORC_MAXSLOT], bc_gp[3];
    double ed_vel_lo[9], ed_vel_hi[9];   /* ext_dir values [n*3+d] for the velocity fill */
    int vlobc[9], vhibc[9];        /* tensor-solve LinOp BC per velocity component [n*3+d] */
    int slobc[3 * ORC_MAXSCAL], shibc[3 * ORC_MAXSCAL];   /* scalar-diffusion LinOp BC [n*3+d] per scalar slot */
    /* ---- AMR (orc_amr.c) ---- */
    int level, ratio;              /* ratio to the next coarser level */
    struct orc_ns_state *crse, *fine;
    int nbox; int* boxes;          /* level > 0: the level's boxes (6 ints each) */
    orc_fab cov;                   /* cell, 0 ghost: 1 on the level's cells (p == NULL on level 0) */
    double st_new, st_old;         /* StateData times of State_Type (Point) */
    double pt_new[2], pt_old[2];   /* time intervals of Press_Type / Gradp_Type (Interval): NavierStokesBase::setTimeLevel, NavierStokesBase.cpp:2978-2996 */
    int iteration, ncycle;         /* of the advance in progress */
    double stop_time;              /* single-level driver (orc_ns_step): computeNewDt's stop_time clamp */
    orc_fab mac_phi;               /* MacProj::mac_phi_crse[level] */
    orc_fab rho_avg, p_avg;        /* level > 0 */
    orc_fab Vsync, Ssync;          /* level < finest: 3 / nstate-3 comps, 1 ghost */
    orc_fab reg_adv[3], reg_visc[3], reg_mac[3];   /* level > 0: coarse-level faces, nstate / nstate / 1 comps */
    orc_fab sync_reg;              /* level > 0: coarse-level nodes (single-valued restatement, orc_amr.c) */
    struct orc_syncreg* sync_lit;  /* level > 0: the literal box-by-box SyncRegister (orc_syncreg.c) */
    orc_fab sync_resid_crse;       /* scratch of the last level projection (coarse-level nodes), see ns_level_project */
};

#define S_NEW(s) (&(s)->S[(s)->inew])
#define S_OLD(s) (&(s)->S[1 - (s)->inew])
#define P_NEW(s) (&(s)->P[(s)->pnew])
#define P_OLD(s) (&(s)->P[1 - (s)->pnew])
#define GP_NEW(s) (&(s)->Gp[(s)->pnew])
#define GP_OLD(s) (&(s)->Gp[1 - (s)->pnew])

/* ---- orc_ns.c ---- */
/* 1 if cell (i,j,k) (periodic images wrapped) belongs to the level */
int ns_covered(const orc_ns_state* s, int i, int j, int k);
/* 1 if cell (i,j,k) lies in grow(box, ng) of one of the level's boxes or their periodic images (level 0: in the domain grown by ng) */
int ns_in_grown(const orc_ns_state* s, int i, int j, int k, int ng);
/* AmrLevel::FillPatch of State_Type (type 0) / Gradp_Type (type 1) at `time`: whole-domain fab with ng filled ghost cells */
orc_fab ns_fillpatch_time(const orc_ns_state* s, double time, int type, int sc, int nc, int ng);
void ns_set_time_level(orc_ns_state* s, double time, double dt_old, double dt_new);
void ns_reset_state(orc_ns_state* s, double time, double dt_old, double dt_new);
double ns_advance(orc_ns_state* s, double dt, int iteration, int ncycle);
double ns_est_time_step(orc_ns_state* s);
void ns_make_rho_curr_time(orc_ns_state* s);
void ns_fill_gp(orc_ns_state* s, orc_fab* G, double time);
void ns_set_outflow_bcs(const orc_ns_state* s, orc_fab* phi, const orc_fab* rho);
/* have_divu: calc_divu into the new (which = 0) or old (1) Divu component, calc_dsdt into the new Dsdt; divu at the old time + dt/2 dsdt */
void ns_calc_divu(orc_ns_state* s, int use_new);
void ns_calc_dsdt(orc_ns_state* s, double dt);
orc_fab ns_divu_half(const orc_ns_state* s, double dt, int ng, int with_dsdt);
void ns_set_inflow_ghosts(const orc_ns_state* s, orc_fab* vel, double inflow_scale);
/* NavierStokes::getViscTerms at the time of Sdata (S_OLD or S_NEW of the level), 1 filled ghost cell */
void ns_get_visc_terms_vel(const orc_ns_state* s, orc_fab* visc /*3 comps, 1 ghost*/, const orc_fab* Sdata);
void ns_get_visc_terms_scalar(const orc_ns_state* s, orc_fab* visc /*1 comp, 1 ghost*/, const orc_fab* Sdata, int comp);
void ns_scalar_level(const orc_ns_state* s, orc_abec_level* L, int comp, double alpha, double beta, const orc_fab* a);

/* ---- flux registers of the interface between level s (fine) and s->crse (orc_amr.c) ---- */
/* side of coarse face f (direction d): 0 not a coarse/fine face, +1 the fine level is on the low side (the coarse cell outside is
 * the one at index f), -1 the fine level is on the high side (outside cell f-1) */
int reg_side(const orc_ns_state* fine, int d, int i, int j, int k);
void reg_setval(orc_fab reg[3], double v);
void reg_crse_init(const orc_ns_state* fine, orc_fab reg[3], const orc_fab* flux /*coarse faces*/, int d, int sc, int dc, int nc, double mult, int add);
void reg_fine_add(const orc_ns_state* fine, orc_fab reg[3], const orc_fab* flux /*fine faces*/, int d, int sc, int dc, int nc, double mult);
void reg_reflux(const orc_ns_state* fine, orc_fab reg[3], orc_fab* S /*coarse cells*/, double volume, double scale, int sc, int dc, int nc);

/* ---- sync registers and residuals (orc_amr.c) ---- */
/* sync residual of level s towards its finer level (Hydro::NodalProjector::computeSyncResidualCoarse): nodal fab of level s */
orc_fab amr_sync_resid_crse(const orc_ns_state* s, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig, const orc_fab* rhcc);
/* sync residual of level s (> 0) towards its coarser level (computeSyncResidualFine): nodal fab of level s */
orc_fab amr_sync_resid_fine(const orc_ns_state* s, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig, const orc_fab* rhcc);
void syncreg_crse_init(orc_ns_state* fine, const orc_fab* resid_crse /*nodes of fine->crse*/, double mult);   /* SyncRegister::CrseInit */
void syncreg_fine_add(orc_ns_state* fine, const orc_fab* resid_fine /*nodes of fine*/, double mult);          /* SyncRegister::FineAdd */

/* ---- literal SyncRegister and box-by-box sync residuals (orc_syncreg.c) ---- */
typedef struct orc_ndmf orc_ndmf;
typedef struct orc_syncreg orc_syncreg;
orc_ndmf* orc_ndmf_create(int nbox, const int* boxes, int ng);
void orc_ndmf_destroy(orc_ndmf* m);
orc_syncreg* orc_syncreg_create(int nbox, const int* fine_boxes, int ratio);
void orc_syncreg_destroy(orc_syncreg* sr);
void orc_syncreg_setval(orc_syncreg* sr, double v);
void orc_syncreg_crse_init(orc_syncreg* sr, orc_ndmf* resid_crse, const orc_geom* cgeom, double mult);
void orc_syncreg_fine_add(orc_syncreg* sr, orc_ndmf* resid_fine, const orc_geom* cgeom, double mult);
void orc_syncreg_comp_add(orc_syncreg* sr, orc_ndmf* resid_fine, const orc_geom* fgeom, const orc_geom* cgeom, int nP, const int* Pboxes, double mult);
void orc_syncreg_init_rhs(orc_syncreg* sr, orc_ndmf* rhs, const orc_geom* geom, const int phys_lo[3], const int phys_hi[3]);
orc_ndmf* orc_sync_resid_fine_boxes(const orc_ns_state* s, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig, const orc_fab* rhcc);
orc_ndmf* orc_sync_resid_crse_boxes(const orc_ns_state* s, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig, const orc_fab* rhcc);
orc_ndmf* orc_level_ndmf(const orc_ns_state* s, int ng);
void orc_ndmf_to_domain(const orc_ndmf* m, orc_fab* out);
/* 1 (default): the multi-level step feeds MLsyncProject from the literal register; 0: from the single-valued restatement.  Both are
 * always run; orc_syncreg_last_diff() = max |difference| of the two right-hand sides of the last MLsyncProject (nodes strictly inside
 * the fine level excluded: the literal 3-D register does not mask them) relative to max |rhs| */
extern int orc_syncreg_literal;
extern double orc_syncreg_diff_max;

/* ---- nodal pieces shared with the composite solver (orc_nodal.c) ---- */
void orc_nodal_fill_bc(const orc_geom* g, orc_fab* x, const int lobc[3], const int hibc[3]);
void orc_sigma_fill_bc(const orc_geom* g, orc_fab* s);
void orc_nodal_divu_bc(const orc_geom* g, orc_fab* rhs, const orc_fab* vel, const int lobc[3], const int hibc[3]);

/* ---- operators on refined levels (orc_tensor.c, orc_abec.c) ---- */
void orc_tensor_apply_cf(const orc_geom* g, int nbox, const int* boxes, int ratio, orc_fab* y, orc_fab* u, double alpha, double beta,
                         const orc_fab* a, orc_fab* const eta[3], const int* lobc, const int* hibc, int maxorder, const orc_fab* cvel);
void orc_tensor_solve_cf(const orc_geom* g, int nbox, const int* boxes, int ratio, orc_fab* u, const orc_fab* rhs, double alpha, double beta,
                         const orc_fab* a, orc_fab* const eta[3], const int* lobc, const int* hibc, const orc_fab* cvel,
                         double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);
void orc_tensor_extensive_flux(const orc_geom* g, int nbox, const int* boxes, int ratio, orc_fab* flux[3], const orc_fab* u, orc_fab* const eta[3],
                               double fac, int add, const orc_fab* cvel, int maxorder);
void orc_abec_extensive_flux(const orc_abec_level* L, orc_fab* flux[3], const orc_fab* phi, double fac, int add);
void orc_cf_set_bcval(const orc_fab* b, int inhomog, int maxorder);
void orc_cf_interp_bndry(const orc_abec_level* L, int ratio, const orc_fab* cphi, orc_fab* bcval);
void orc_abec_solve_cf(const orc_abec_level* L, orc_fab* phi, const orc_fab* rhs, const int lobc[3], const int hibc[3],
                       const orc_fab* cf_bcval, double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);

#endif
