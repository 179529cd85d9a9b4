/* oracle/orc_ns.c -- single-level NavierStokes::advance + initialisation sequence restated on the CPU
 * (test infrastructure only; PARITY UNPINNED for the upstream kernels, see orc.h).
 *
 * Follows the IN-TREE orchestration line by line:
 *   NavierStokes::advance                  Source/NavierStokes.cpp:543-691
 *   NSB::advance_setup                     Source/NavierStokesBase.cpp:613-741
 *   NSB::predict_velocity                  Source/NavierStokesBase.cpp:4376-4512
 *   NSB::mac_project / MacProj::mac_project Source/NavierStokesBase.cpp:2070-2109, Source/MacProj.cpp:225-353
 *   NSB::velocity_advection                Source/NavierStokesBase.cpp:3358-3470
 *   NavierStokes::scalar_advection         Source/NavierStokes.cpp:698-812
 *   NSB::scalar_advection_update           Source/NavierStokesBase.cpp:2730-2972
 *   NSB::velocity_advection_update         Source/NavierStokesBase.cpp:3523-3655
 *   NSB::initial_velocity_diffusion_update Source/NavierStokesBase.cpp:3658-3749
 *   Diffusion::diffuse_tensor_velocity     Source/Diffusion.cpp:650-957
 *   Diffusion::getTensorViscTerms          Source/Diffusion.cpp:1655-1777
 *   Projection::level_project              Source/Projection.cpp:166-450
 *   Projection::initialVelocityProject     Source/Projection.cpp:615-838
 *   Projection::initialSyncProject         Source/Projection.cpp:970-1185
 *   NavierStokes::post_init / post_init_press Source/NavierStokes.cpp:1254-1432
 *   NSB::estTimeStep / computeNewDt        Source/NavierStokesBase.cpp:1353-1510, 945-1035
 *   NavierStokes::scalar_diffusion_update  Source/NavierStokes.cpp:867-1000 -> Diffusion::diffuse_scalar Source/Diffusion.cpp:207-599
 *   Diffusion::getViscTerms (scalars)      Source/Diffusion.cpp:1539-1652
 *   physical BC tables                     Source/NS_BC.H:7-55, Source/NS_setup.cpp:21-128, Source/NS_bcfill.H:17-180
 * Scope: one level; each direction periodic or bounded by SlipWall / NoSlipWall (moving walls through
 * xlo.velocity ... zhi.velocity); constant viscosity / tracer diffusivity, no divu; nstate = 5 + do_trac2 + do_temp
 * (u,v,w,rho,tracer), do_mom_diff = 0 or 1, Godunov_PLM.
 */
#include "orc_ns_int.h"

void orc_ns_default_params(orc_ns_params* p)
{
    p->cfl = 0.8; p->visc_coef = 0.0; p->be_cn_theta = 0.5; p->gravity = 0.0;
    p->mac_tol = 1.e-12; p->mac_abs_tol = 1.e-16; p->proj_tol = 1.e-12; p->proj_abs_tol = 1.e-16; p->visc_tol = 1.e-10;
    p->use_forces_in_trans = 0; p->do_mom_diff = 0; p->init_iter = 2; p->init_vel_iter = 1;
    p->init_shrink = 1.0; p->change_max = 1.1; p->fixed_dt = -1.0; p->nscal = 2; p->verbose = 0;
    p->init_dt = -1.0; p->tracer_diff_coef = 0.0;
    for (int d = 0; d < 3; ++d) p->phys_lo[d] = p->phys_hi[d] = 0;
    for (int q = 0; q < 9; ++q) p->wall_vel_lo[q] = p->wall_vel_hi[q] = 0.0;
    for (int q = 0; q < 12; ++q) p->scal_bc_lo[q] = p->scal_bc_hi[q] = 0.0;
    p->do_cons_trac = 0;
    p->do_denminmax = 0; p->do_scalminmax = 0;
    p->do_trac2 = 0; p->do_cons_trac2 = 0; p->tracer2_diff_coef = 0.0; p->do_temp = 0; p->temp_cond_coef = 0.0;
    p->use_ppm = 0;
}

/* BCType of a velocity component / scalar / grad p component for a physical BC (Source/NS_BC.H:7-35) */
enum { PHYS_INTERIOR = 0, PHYS_INFLOW = 1, PHYS_OUTFLOW = 2, PHYS_SYMMETRY = 3, PHYS_SLIPWALL = 4, PHYS_NOSLIPWALL = 5 };
static int vel_bctype(int phys, int normal)
{
    if (phys == PHYS_INTERIOR) return ORC_BC_INT_DIR;
    if (phys == PHYS_INFLOW) return ORC_BC_EXT_DIR;
    if (phys == PHYS_OUTFLOW) return ORC_BC_FOEXTRAP;
    if (phys == PHYS_SYMMETRY) return normal ? ORC_BC_REFLECT_ODD : ORC_BC_REFLECT_EVEN;
    if (phys == PHYS_NOSLIPWALL) return ORC_BC_EXT_DIR;
    return normal ? ORC_BC_EXT_DIR : ORC_BC_HOEXTRAP;          /* SlipWall */
}
static int scal_bctype(int phys)
{
    if (phys == PHYS_INTERIOR) return ORC_BC_INT_DIR;
    if (phys == PHYS_SYMMETRY) return ORC_BC_REFLECT_EVEN;
    return phys == PHYS_INFLOW ? ORC_BC_EXT_DIR : ORC_BC_FOEXTRAP;
}
static int temp_bctype(int phys)             /* temp_bc, Source/NS_BC.H:37-40 */
{
    if (phys == PHYS_INTERIOR) return ORC_BC_INT_DIR;
    if (phys == PHYS_INFLOW) return ORC_BC_EXT_DIR;
    return phys == PHYS_OUTFLOW ? ORC_BC_HOEXTRAP : ORC_BC_REFLECT_EVEN;
}
static int gp_bctype(int phys, int normal)   /* norm/tang_gradp_bc */
{
    if (phys == PHYS_INTERIOR) return ORC_BC_INT_DIR;
    if (phys == PHYS_SYMMETRY) return normal ? ORC_BC_REFLECT_ODD : ORC_BC_REFLECT_EVEN;
    return ORC_BC_FOEXTRAP;
}
static int phys_ok(int phys) { return phys == PHYS_INFLOW || phys == PHYS_OUTFLOW || phys == PHYS_SYMMETRY || phys == PHYS_SLIPWALL || phys == PHYS_NOSLIPWALL; }
/* Diffusion::setDomainBC, Source/Diffusion.cpp:1886-1941 */
static int linop_of_bctype(int bct)
{
    if (bct == ORC_BC_EXT_DIR) return ORC_LO_DIRICHLET;
    if (bct == ORC_BC_FOEXTRAP || bct == ORC_BC_HOEXTRAP || bct == ORC_BC_REFLECT_EVEN) return ORC_LO_NEUMANN;
    if (bct == ORC_BC_REFLECT_ODD) return ORC_LO_REFLECT_ODD;
    return ORC_LO_PERIODIC;
}

orc_ns_state* orc_ns_create(const orc_geom* g, const orc_ns_params* p, const orc_mg_opts* o)
{
    orc_ns_state* s = (orc_ns_state*)calloc(1, sizeof(orc_ns_state));
    s->g = *g; s->p = *p; s->o = *o;
    /* NavierStokes::Initialize (NavierStokes.cpp:43-55): Density, Tracer, [Tracer2], [Temp] */
    s->nstate = Tracer + 1; s->Tracer2 = -1; s->Temp = -1;
    if (p->do_trac2) s->Tracer2 = s->nstate++;
    if (p->do_temp) s->Temp = s->nstate++;
    s->nscal = s->nstate - Density;
    for (int n = 0; n < ORC_MAXSCAL; ++n) { s->scal_cons[n] = 0; s->scal_rho_flag[n] = 1; s->scal_diff[n] = 0.0; }
    s->scal_cons[0] = 1; s->scal_diff[0] = -1.0;                                  /* density: conservative, never diffusive (NS_setup.cpp:303, NavierStokes.cpp:291) */
    s->scal_cons[1] = p->do_cons_trac != 0; s->scal_rho_flag[1] = p->do_cons_trac ? 2 : 0; s->scal_diff[1] = p->tracer_diff_coef;   /* NS_setup.cpp:304-310 */
    if (p->do_trac2) { const int n = s->Tracer2 - Density; s->scal_cons[n] = p->do_cons_trac2 != 0; s->scal_rho_flag[n] = p->do_cons_trac2 ? 2 : 0; s->scal_diff[n] = p->tracer2_diff_coef; }
    if (p->do_temp) { const int n = s->Temp - Density; s->scal_cons[n] = 0; s->scal_rho_flag[n] = 1; s->scal_diff[n] = p->temp_cond_coef; }   /* NS_setup.cpp:302, default RhoInverse_Laplacian_S */
    s->have_divu = p->do_temp != 0; s->Divu = s->nstate; s->Dsdt = s->nstate + 1; s->nalloc = s->nstate + (s->have_divu ? 2 : 0);
    for (int q = 0; q < 2; ++q) {
        s->S[q] = orc_alloc(g->n, ORC_CELL, 1, s->nalloc);
        s->P[q] = orc_alloc(g->n, ORC_NODE, 1, 1);
        s->Gp[q] = orc_alloc(g->n, ORC_CELL, 1, 3);
    }
    for (int d = 0; d < 3; ++d) { s->umac[d] = orc_alloc(g->n, ORC_FACE[d], 1, 1); orc_setval(&s->umac[d], 1.e40); }
    s->aofs = orc_alloc(g->n, ORC_CELL, 0, s->nstate);
    s->rho_ptime = orc_alloc(g->n, ORC_CELL, 1, 1);
    s->rho_ctime = orc_alloc(g->n, ORC_CELL, 1, 1);
    s->rho_half = orc_alloc(g->n, ORC_CELL, 1, 1);
    s->mac_phi = orc_alloc(g->n, ORC_CELL, 1, 1);
    s->level = 0; s->ratio = 1; s->crse = s->fine = NULL; s->nbox = 0; s->boxes = NULL; s->cov.p = NULL;
    s->iteration = 1; s->ncycle = 1;
    ns_set_time_level(s, 0.0, 0.0, 0.0);
    s->stop_time = -1.0;
    for (int d = 0; d < 3; ++d) {
        const int plo = g->periodic[d] ? PHYS_INTERIOR : p->phys_lo[d], phi_ = g->periodic[d] ? PHYS_INTERIOR : p->phys_hi[d];
        if (!g->periodic[d] && !(phys_ok(plo) && phys_ok(phi_))) {
            fprintf(stderr, "orc_ns_create: non-periodic direction %d needs Inflow(1)/Outflow(2)/Symmetry(3)/SlipWall(4)/NoSlipWall(5) on both sides\n", d);
            free(s); return NULL;
        }
        /* MacProj::set_mac_solve_bc (Source/MacProj.cpp:1187-1208): outflow Dirichlet, everything else Neumann;
         * Projection.cpp:2434-2464: outflow Dirichlet, inflow "inflow", everything else Neumann */
        s->lobc[d] = g->periodic[d] ? ORC_LO_PERIODIC : (plo == PHYS_OUTFLOW ? ORC_LO_DIRICHLET : ORC_LO_NEUMANN);
        s->hibc[d] = g->periodic[d] ? ORC_LO_PERIODIC : (phi_ == PHYS_OUTFLOW ? ORC_LO_DIRICHLET : ORC_LO_NEUMANN);
        s->nlobc[d] = (!g->periodic[d] && plo == PHYS_INFLOW) ? ORC_LO_INFLOW : s->lobc[d];
        s->nhibc[d] = (!g->periodic[d] && phi_ == PHYS_INFLOW) ? ORC_LO_INFLOW : s->hibc[d];
        for (int n = 0; n < 3; ++n) {
            s->bc_vel[n].lo[d] = vel_bctype(plo, n == d); s->bc_vel[n].hi[d] = vel_bctype(phi_, n == d);
            s->bc_gp[n].lo[d] = gp_bctype(plo, n == d); s->bc_gp[n].hi[d] = gp_bctype(phi_, n == d);
            s->ed_vel_lo[n * 3 + d] = p->wall_vel_lo[d * 3 + n]; s->ed_vel_hi[n * 3 + d] = p->wall_vel_hi[d * 3 + n];
            s->vlobc[n * 3 + d] = linop_of_bctype(s->bc_vel[n].lo[d]); s->vhibc[n * 3 + d] = linop_of_bctype(s->bc_vel[n].hi[d]);
        }
        for (int n = 0; n < s->nscal; ++n) {
            const int is_temp = Density + n == s->Temp;                            /* set_scalar_bc / set_temp_bc (NS_setup.cpp:263-283) */
            s->bc_scal[n].lo[d] = is_temp ? temp_bctype(plo) : scal_bctype(plo); s->bc_scal[n].hi[d] = is_temp ? temp_bctype(phi_) : scal_bctype(phi_);
            s->ed_scal_lo[n * 3 + d] = p->scal_bc_lo[d * 4 + n]; s->ed_scal_hi[n * 3 + d] = p->scal_bc_hi[d * 4 + n];
            s->slobc[n * 3 + d] = linop_of_bctype(s->bc_scal[n].lo[d]); s->shibc[n * 3 + d] = linop_of_bctype(s->bc_scal[n].hi[d]);
        }
        if (s->have_divu) {     /* divu_bc / dsdt_bc, NS_BC.H:42-50; dsdt's ext_dir faces are filled with zero (homogeneous_bf, NS_setup.cpp:383) */
            const int a = s->nscal, b = s->nscal + 1;
            s->bc_scal[a].lo[d] = plo == PHYS_INTERIOR ? ORC_BC_INT_DIR : ORC_BC_REFLECT_EVEN; s->bc_scal[a].hi[d] = phi_ == PHYS_INTERIOR ? ORC_BC_INT_DIR : ORC_BC_REFLECT_EVEN;
            s->bc_scal[b].lo[d] = plo == PHYS_INTERIOR ? ORC_BC_INT_DIR : ((plo == PHYS_INFLOW || plo == PHYS_OUTFLOW) ? ORC_BC_EXT_DIR : ORC_BC_REFLECT_EVEN);
            s->bc_scal[b].hi[d] = phi_ == PHYS_INTERIOR ? ORC_BC_INT_DIR : ((phi_ == PHYS_INFLOW || phi_ == PHYS_OUTFLOW) ? ORC_BC_EXT_DIR : ORC_BC_REFLECT_EVEN);
            s->ed_scal_lo[a * 3 + d] = s->ed_scal_hi[a * 3 + d] = s->ed_scal_lo[b * 3 + d] = s->ed_scal_hi[b * 3 + d] = 0.0;
        }
    }
    return s;
}

void orc_ns_destroy(orc_ns_state* s)
{
    for (int q = 0; q < 2; ++q) { orc_free(&s->S[q]); orc_free(&s->P[q]); orc_free(&s->Gp[q]); }
    for (int d = 0; d < 3; ++d) orc_free(&s->umac[d]);
    orc_free(&s->aofs); orc_free(&s->rho_ptime); orc_free(&s->rho_ctime); orc_free(&s->rho_half);
    orc_free(&s->mac_phi);
    if (s->cov.p) orc_free(&s->cov);
    if (s->rho_avg.p) orc_free(&s->rho_avg);
    if (s->p_avg.p) orc_free(&s->p_avg);
    if (s->Vsync.p) orc_free(&s->Vsync);
    if (s->Ssync.p) orc_free(&s->Ssync);
    for (int d = 0; d < 3; ++d) { if (s->reg_adv[d].p) orc_free(&s->reg_adv[d]); if (s->reg_visc[d].p) orc_free(&s->reg_visc[d]); if (s->reg_mac[d].p) orc_free(&s->reg_mac[d]); }
    if (s->sync_reg.p) orc_free(&s->sync_reg);
    if (s->sync_lit) { orc_syncreg_destroy(s->sync_lit); s->sync_lit = NULL; }
    if (s->sync_resid_crse.p) orc_free(&s->sync_resid_crse);
    free(s->boxes);
    free(s);
}

orc_fab* orc_ns_fab(orc_ns_state* s, int which)
{
    switch (which) {
    case 0: return S_NEW(s);
    case 1: return S_OLD(s);
    case 2: return P_NEW(s);
    case 3: return P_OLD(s);
    case 4: return GP_NEW(s);
    case 5: return GP_OLD(s);
    case 6: case 7: case 8: return &s->umac[which - 6];
    case 9: return &s->aofs;
    case 10: return &s->mac_phi;
    case 11: return &s->rho_half;
    case 12: return s->Vsync.p ? &s->Vsync : NULL;
    case 13: return s->Ssync.p ? &s->Ssync : NULL;
    case 14: return s->rho_avg.p ? &s->rho_avg : NULL;
    case 15: return s->p_avg.p ? &s->p_avg : NULL;
    case 16: return s->sync_reg.p ? &s->sync_reg : NULL;
    default: return NULL;
    }
}
double orc_ns_time(const orc_ns_state* s) { return s->time; }
double orc_ns_dt(const orc_ns_state* s) { return s->dt; }
void orc_ns_last_stats(const orc_ns_state* s, orc_mg_stats* mac, orc_mg_stats* nodal, orc_mg_stats* visc)
{
    if (mac) *mac = s->st_mac;
    if (nodal) *nodal = s->st_nodal;
    if (visc) *visc = s->st_visc;
}

/* NavierStokes::init_TaylorGreen, reference Source/prob/prob_init.cpp:509-560 */
void orc_ns_init_rayleightaylor(orc_ns_state* s, double rho_1, double rho_2, double tra_1, double tra_2, double pertamp, double interface_width)
{
    const double Pi = 3.14159265358979323846264338327950288;
    orc_fab* S = S_NEW(s);
    orc_setval(S, 0.0);
    const orc_geom* g = &s->g;
    const double Lx = g->dx[0] * g->n[0], Ly = g->dx[1] * g->n[1];
    const double splitz = 0.5 * (g->problo[2] + (g->problo[2] + g->dx[2] * g->n[2]));
    const double ranampl = 2. * (0.6544437533747718 - 0.5);
    const double ranphse1 = 2. * Pi * 0.1556190326530211, ranphse2 = 2. * Pi * 0.4196144025537369;
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        const double x = g->problo[0] + (i + 0.5) * g->dx[0], y = g->problo[1] + (j + 0.5) * g->dx[1], z = g->problo[2] + (k + 0.5) * g->dx[2];
        const double pert = ranampl * sin(2.0 * Pi * x / Lx + ranphse1) * sin(2.0 * Pi * y / Ly + ranphse2);
        const double pertheight = splitz - pertamp * pert;
        A4(S, i, j, k, Density) = rho_1 + ((rho_2 - rho_1) / 2.0) * (1.0 + tanh((z - pertheight) / interface_width));
        A4(S, i, j, k, Tracer) = tra_1 + ((tra_2 - tra_1) / 2.0) * (1.0 + tanh((z - pertheight) / interface_width));
        for (int nt = Tracer + 1; nt < s->nstate; ++nt) A4(S, i, j, k, nt) = 1.0;      /* prob_init.cpp:482-485 */
    }
    orc_setval(P_NEW(s), 0.0); orc_setval(P_OLD(s), 0.0);
    orc_setval(GP_NEW(s), 0.0); orc_setval(GP_OLD(s), 0.0);
    s->time = 0.0; s->nstep = 0;
}

void orc_ns_init_taylorgreen(orc_ns_state* s, double vfac, double a, double b, double c, double rho0)
{
    const orc_geom* g = &s->g;
    const double TwoPi = 2.0 * 3.14159265358979323846264338327950288;
    orc_fab* S = S_NEW(s);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        double x = g->problo[0] + (i + 0.5) * g->dx[0];
        double y = g->problo[1] + (j + 0.5) * g->dx[1];
        double z = g->problo[2] + (k + 0.5) * g->dx[2];
        A4(S, i, j, k, 0) = vfac * sin(a * TwoPi * x) * cos(b * TwoPi * y) * cos(c * TwoPi * z);
        A4(S, i, j, k, 1) = -vfac * cos(a * TwoPi * x) * sin(b * TwoPi * y) * cos(c * TwoPi * z);
        A4(S, i, j, k, 2) = 0.0;
        A4(S, i, j, k, Density) = rho0;
        A4(S, i, j, k, Tracer) = (rho0 * vfac * vfac / 16.0) * (2.0 + cos(2.0 * c * TwoPi * z)) * (cos(2.0 * a * TwoPi * x) + cos(2.0 * b * TwoPi * y));
        for (int nt = Tracer + 1; nt < s->nstate; ++nt) A4(S, i, j, k, nt) = 1.0;      /* prob_init.cpp:555-558 */
    }
    orc_setval(P_NEW(s), 0.0); orc_setval(P_OLD(s), 0.0);
    orc_setval(GP_NEW(s), 0.0); orc_setval(GP_OLD(s), 0.0);
    s->time = 0.0; s->nstep = 0;
}

/* probtype 1 (LidDrivenCavity): fluid at rest, rho = rho0, tracer = 0 (reference Source/prob/prob_init.cpp:102-109) */
void orc_ns_init_rest(orc_ns_state* s, double rho0)
{
    orc_fab* S = S_NEW(s);
    orc_setval(S, 0.0);
    const orc_geom* g = &s->g;
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(S, i, j, k, Density) = rho0;
    orc_setval(P_NEW(s), 0.0); orc_setval(P_OLD(s), 0.0);
    orc_setval(GP_NEW(s), 0.0); orc_setval(GP_OLD(s), 0.0);
    s->time = 0.0; s->nstep = 0;
}

/* ---- AMR-aware pieces (a level > 0 is a whole-domain array + coverage mask, see orc_ns_int.h) ---------------------------- */
static inline int wrapi(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }

int ns_covered(const orc_ns_state* s, int i, int j, int k)
{
    int c[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) {
        if (c[d] < 0 || c[d] >= s->g.n[d]) { if (!s->g.periodic[d]) return 0; c[d] = wrapi(c[d], s->g.n[d]); }
    }
    return s->cov.p ? A4(&s->cov, c[0], c[1], c[2], 0) != 0.0 : 1;
}

int ns_in_grown(const orc_ns_state* s, int i, int j, int k, int ng)
{
    const int c[3] = {i, j, k};
    if (s->level == 0) {
        for (int d = 0; d < 3; ++d) if (c[d] < -ng || c[d] > s->g.n[d] - 1 + ng) return 0;
        return 1;
    }
    for (int b = 0; b < s->nbox; ++b) {
        const int* bx = s->boxes + 6 * b;
        int in = 1;
        for (int d = 0; d < 3 && in; ++d) {
            int ok = 0;
            for (int sh = -1; sh <= 1 && !ok; ++sh) {
                if (sh != 0 && !s->g.periodic[d]) continue;
                const int q = c[d] - sh * s->g.n[d];
                if (q >= bx[d] - ng && q <= bx[3 + d] + ng) ok = 1;
            }
            in = ok;
        }
        if (in) return 1;
    }
    return 0;
}

/* NavierStokesBase::setTimeLevel (NavierStokesBase.cpp:2978-2996) with amrex::StateData::setTimeLevel semantics: State_Type is a
 * Point type, Press_Type / Gradp_Type are Interval types (NS_setup.cpp:228-341) shifted back by dt_old */
void ns_set_time_level(orc_ns_state* s, double time, double dt_old, double dt_new)
{
    (void)dt_new;
    s->st_new = time; s->st_old = time - dt_old;
    const double tp = time - dt_old;               /* state[Press_Type].setTimeLevel(time-dt_old,dt_old,dt_old) */
    s->pt_new[0] = tp; s->pt_new[1] = tp + dt_old;
    s->pt_old[0] = tp - dt_old; s->pt_old[1] = tp;
}

/* StateData::swapTimeLevels(dt) */
static void swap_time_levels(orc_ns_state* s, double dt)
{
    s->st_old = s->st_new; s->st_new += dt;
    s->pt_old[0] = s->pt_new[0]; s->pt_old[1] = s->pt_new[1];
    s->pt_new[0] = s->pt_new[1]; s->pt_new[1] += dt;
}

/* the level's own State_Type data at `time` (valid cells only matter): amrex::StateData::getData for a Point type -- old or new
 * within 1e-3 (t_new - t_old) of their times, linear interpolation otherwise.  Returns a 0-ghost fab with nc comps. */
static orc_fab state_at(const orc_ns_state* s, double time, int sc, int nc)
{
    const orc_geom* g = &s->g;
    orc_fab f = orc_alloc(g->n, ORC_CELL, 0, nc);
    const orc_fab *So = S_OLD(s), *Sn = S_NEW(s);
    const double teps = 1.e-3 * fabs(s->st_new - s->st_old);
    double a = 0.0, b = 1.0;
    if (fabs(time - s->st_new) <= teps) { a = 0.0; b = 1.0; }
    else if (fabs(time - s->st_old) <= teps) { a = 1.0; b = 0.0; }
    else { a = (s->st_new - time) / (s->st_new - s->st_old); b = (time - s->st_old) / (s->st_new - s->st_old); }
    for (int n = 0; n < nc; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        if (a == 0.0) A4(&f, i, j, k, n) = A4(Sn, i, j, k, sc + n);
        else if (b == 0.0) A4(&f, i, j, k, n) = A4(So, i, j, k, sc + n);
        else A4(&f, i, j, k, n) = a * A4(So, i, j, k, sc + n) + b * A4(Sn, i, j, k, sc + n);
    }
    return f;
}

/* Gradp_Type (Interval): the data whose time interval contains `time` (amrex::StateData::getData) */
static orc_fab gradp_at(const orc_ns_state* s, double time)
{
    const orc_geom* g = &s->g;
    const double teps = 1.e-3 * fabs(s->pt_new[0] - s->pt_old[0]);
    const orc_fab* G;
    if (time >= s->pt_new[0] - teps && time <= s->pt_new[1] + teps) G = GP_NEW(s);
    else if (time >= s->pt_old[0] - teps && time <= s->pt_old[1] + teps) G = GP_OLD(s);
    else { fprintf(stderr, "orc gradp_at: level %d has no Gradp data at time %.17g (old [%g,%g] new [%g,%g])\n", s->level, time, s->pt_old[0], s->pt_old[1], s->pt_new[0], s->pt_new[1]); abort(); }
    orc_fab f = orc_alloc(g->n, ORC_CELL, 0, 3);
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&f, i, j, k, n) = A4(G, i, j, k, n);
    return f;
}

/* AmrLevel::FillPatch (FillPatchSingleLevel / FillPatchTwoLevels with cell_cons_interp, NS_setup.cpp:206-394): the level's own data
 * where the level has cells, conservative-linear interpolation of the (recursively FillPatch'ed) coarse data elsewhere, periodic
 * images, then the physical BC.  type 0: State_Type comps [sc, sc+nc); type 1: Gradp_Type (3 comps). */
orc_fab ns_fillpatch_time(const orc_ns_state* s, double time, int type, int sc, int nc, int ng)
{
    const orc_geom* g = &s->g;
    const orc_bcrec* bc = type == 1 ? s->bc_gp : (sc < 3 ? s->bc_vel + sc : s->bc_scal + (sc - 3));
    const double* edlo = type == 1 ? NULL : (sc < 3 ? s->ed_vel_lo + 3 * sc : s->ed_scal_lo + 3 * (sc - 3));
    const double* edhi = type == 1 ? NULL : (sc < 3 ? s->ed_vel_hi + 3 * sc : s->ed_scal_hi + 3 * (sc - 3));
    orc_fab own = type == 1 ? gradp_at(s, time) : state_at(s, time, sc, nc);
    orc_fab f = orc_alloc(g->n, ORC_CELL, ng, nc);
    if (s->level > 0) {
        const int r = s->ratio;
        const int ngc = (ng + r - 1) / r + 2;          /* coarsen(ghost region) + 1 for the slopes + 1 for the one-sided slope form */
        orc_fab c = ns_fillpatch_time(s->crse, time, type, sc, nc, ngc);
        const int cdomlo[3] = {0, 0, 0}, cdomhi[3] = {s->crse->g.n[0] - 1, s->crse->g.n[1] - 1, s->crse->g.n[2] - 1};
        const int vlo[3] = {1, 1, 1}, vhi[3] = {0, 0, 0};      /* empty "valid box": interpolate everywhere, the level's cells are overwritten below */
        orc_fill_coarse_fine(&f, f.lo, f.hi, vlo, vhi, &c, cdomlo, cdomhi, g->periodic, r, bc);
        orc_free(&c);
    }
    for (int n = 0; n < nc; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        if (s->level == 0 || A4(&s->cov, i, j, k, 0) != 0.0) A4(&f, i, j, k, n) = A4(&own, i, j, k, n);
    orc_free(&own);
    orc_fill_periodic(&f, g, ORC_CELL);
    orc_fill_physbc_cc(&f, g, bc, edlo, edhi);
    return f;
}

/* FillPatch of comps [sc, sc+nc) of src (S_OLD or S_NEW of the level) into a fresh fab with ng ghosts */
static orc_fab fillpatch(const orc_ns_state* s, const orc_fab* src, int sc, int nc, int ng, const orc_bcrec* bc)
{
    if (s->level > 0) {
        if (src != S_OLD(s) && src != S_NEW(s)) { fprintf(stderr, "orc fillpatch: level > 0 needs a StateData source\n"); abort(); }
        return ns_fillpatch_time(s, src == S_OLD(s) ? s->st_old : s->st_new, 0, sc, nc, ng);
    }
    const int is_vel = (bc == s->bc_vel);
    const orc_geom* g = &s->g;
    orc_fab f = orc_alloc(g->n, ORC_CELL, ng, nc);
    for (int n = 0; n < nc; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(&f, i, j, k, n) = A4(src, i, j, k, sc + n);
    orc_fill_periodic(&f, g, ORC_CELL);
    const int is_scal = (bc >= s->bc_scal && bc < s->bc_scal + ORC_MAXSLOT);
    const long so = is_scal ? 3 * (bc - s->bc_scal) : 0;
    if (bc) orc_fill_physbc_cc(&f, g, bc, is_vel ? s->ed_vel_lo : (is_scal ? s->ed_scal_lo + so : NULL),
                               is_vel ? s->ed_vel_hi : (is_scal ? s->ed_scal_hi + so : NULL));
    return f;
}

/* FillPatch(Gradp_Type) of array G (GP_OLD or GP_NEW) at `time`: ghost cells only (Projection.cpp:2564-2565,
 * NavierStokesBase.cpp:4421-4422); the valid data are left alone */
void ns_fill_gp(orc_ns_state* s, orc_fab* G, double time)
{
    const orc_geom* g = &s->g;
    if (s->level == 0) {
        orc_fill_periodic(G, g, ORC_CELL);
        orc_fill_physbc_cc(G, g, s->bc_gp, NULL, NULL);
        return;
    }
    orc_fab f = ns_fillpatch_time(s, time, 1, 0, 3, 1);
    for (int n = 0; n < 3; ++n)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
        const int inside = i >= 0 && i < g->n[0] && j >= 0 && j < g->n[1] && k >= 0 && k < g->n[2];
        if (inside && A4(&s->cov, i, j, k, 0) != 0.0) continue;
        A4(G, i, j, k, n) = A4(&f, i, j, k, n);
    }
    orc_free(&f);
}
static void fill_ghosts(const orc_ns_state* s, orc_fab* f, const int type[3])
{
    orc_fill_periodic(f, &s->g, type);
}

/* test hook: scale factor applied to the extrapolated wall ghost cells of the viscous terms (1 = the algorithm) */
static double s_extrap_scale = 1.0;
void orc_ns_test_set_extrap_scale(double v) { s_extrap_scale = v; }

static void floor_small(orc_fab* f)
{
    size_t N = orc_npts(f) * (size_t)f->nc;
    for (size_t q = 0; q < N; ++q) if (fabs(f->p[q]) <= 1.e-20) f->p[q] = 0.0;
}

static int is_diffusive_vel(const orc_ns_state* s) { return s->p.visc_coef > 0.0; }

static void make_eta(const orc_ns_state* s, orc_fab eta[3])
{
    for (int d = 0; d < 3; ++d) { eta[d] = orc_alloc(s->g.n, ORC_FACE[d], 0, 1); orc_setval(&eta[d], s->p.visc_coef); }
}

/* Extrapolater::FirstOrderExtrap role (reference Source/NavierStokes.cpp:2047): give the ghost cells outside the physical
 * domain a first-order value.  Restated as: copy of the nearest cell inside the domain (index clamp in the non-periodic
 * directions).  These cells only feed Godunov states on wall faces that the wall BC overrides, see
 * tests/test_cpu_oracle.py::test_wall_ghost_forcing_is_immaterial. */
static void first_order_extrap(orc_fab* f, const orc_geom* g)
{
    for (int n = 0; n < f->nc; ++n)
    for (int k = f->lo[2]; k <= f->hi[2]; ++k) for (int j = f->lo[1]; j <= f->hi[1]; ++j) for (int i = f->lo[0]; i <= f->hi[0]; ++i) {
        int q[3] = {i, j, k}, out = 0;
        for (int d = 0; d < 3; ++d) {
            if (g->periodic[d]) continue;
            if (q[d] < 0) { q[d] = 0; out = 1; } else if (q[d] > g->n[d] - 1) { q[d] = g->n[d] - 1; out = 1; }
        }
        if (out) A4(f, i, j, k, n) = A4(f, q[0], q[1], q[2], n) * s_extrap_scale;
    }
}

static int is_diffusive_scal(const orc_ns_state* s, int comp) { return s->scal_diff[comp - Density] > 0.0; }

/* ---- refined levels: coarse data of the coarse/fine boundary conditions and the C/F part of FirstOrderExtrap ---- */
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
/* the coarse level's data at time t (FillPatch of the coarse level, 1 ghost cell): the crsedata of Diffusion.cpp:733-744, 1725-1736 */
static orc_fab crse_vel_at(const orc_ns_state* s, double t) { return ns_fillpatch_time(s->crse, t, 0, Xvel, 3, 1); }
static orc_fab crse_scalar_at(const orc_ns_state* s, double t, int comp, int over_rho)
{
    orc_fab c = ns_fillpatch_time(s->crse, t, 0, comp, 1, 1);
    if (over_rho) {
        orc_fab r = ns_fillpatch_time(s->crse, t, 0, Density, 1, 1);
        const size_t N = orc_npts(&c);
        for (size_t q = 0; q < N; ++q) c.p[q] /= r.p[q];
        orc_free(&r);
    }
    return c;
}
static double time_of(const orc_ns_state* s, const orc_fab* Sdata) { return Sdata == S_OLD(s) ? s->st_old : s->st_new; }
/* ghost cells at coarse/fine boundaries (cells of the domain outside the level): the mean of the level's cells among the face
 * neighbours, else among the edge neighbours, else among the corner neighbours (the product's single-valued rule, navierstokes.hip) */
static void first_order_extrap_cf(const orc_ns_state* s, orc_fab* f)
{
    if (s->level == 0) return;
    const orc_geom* g = &s->g;
    orc_fab src = orc_alloc(g->n, ORC_CELL, 1, f->nc);
    orc_copy_all(&src, f);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        if (A4(&s->cov, i, j, k, 0) != 0.0) continue;
        for (int cls = 1; cls <= 3; ++cls) {
            int cnt = 0;
            double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                if ((dx != 0) + (dy != 0) + (dz != 0) != cls) continue;
                if (!ns_covered(s, i + dx, j + dy, k + dz)) continue;
                int q[3] = {i + dx, j + dy, k + dz};
                for (int d = 0; d < 3; ++d) if (g->periodic[d]) q[d] = (q[d] % g->n[d] + g->n[d]) % g->n[d];
                ++cnt;
                for (int n = 0; n < f->nc; ++n) sum[n] += A4(&src, q[0], q[1], q[2], n);
            }
            if (cnt > 0) { for (int n = 0; n < f->nc; ++n) A4(f, i, j, k, n) = sum[n] / (double)cnt; break; }
        }
    }
    orc_free(&src);
    orc_fill_periodic(f, g, ORC_CELL);
}

/* the (constant-coefficient) scalar diffusion operator of state component comp: MLABecLaplacian with b = diffusivity on faces */
void ns_scalar_level(const orc_ns_state* s, orc_abec_level* L, int comp, double alpha, double beta, const orc_fab* a)
{
    memset(L, 0, sizeof(*L));
    L->g = s->g; L->alpha = alpha; L->beta = beta; L->ncomp = 1; L->tensor = 0;
    if (a) L->a = *a; else L->a.p = NULL;
    for (int d = 0; d < 3; ++d) { L->b[d] = orc_alloc(s->g.n, ORC_FACE[d], 0, 1); orc_setval(&L->b[d], s->scal_diff[comp - Density]); }
}

/* NavierStokes::getViscTerms for one scalar (Diffusion::getViscTerms, Diffusion.cpp:1540-1650): visc = div(beta grad S(time)),
 * rho_flag 2 (Laplacian_SoverRho): of S / rho; called at the old time only (Sdata = S_old, get_rho(time) = rho_ptime) */
void ns_get_visc_terms_scalar(const orc_ns_state* s, orc_fab* visc /*1 comp, 1 ghost*/, const orc_fab* Sdata, int comp)
{
    const orc_geom* g = &s->g;
    const int sn = comp - Density, over_rho = s->scal_rho_flag[sn] == 2;
    const int* slobc = s->slobc + 3 * sn; const int* shibc = s->shibc + 3 * sn;
    orc_setval(visc, 1.e40);
    if (!is_diffusive_scal(s, comp)) { orc_setval(visc, 0.0); return; }
    orc_fab stmp = fillpatch(s, Sdata, comp, 1, 1, &s->bc_scal[sn]);
    if (over_rho) {             /* rho_flag 2 (Diffusion.cpp:1612-1615): evaluate div beta grad(S/rho), rho = get_rho(time) incl. the ghost cells */
        const size_t N = orc_npts(&stmp);
        for (size_t q = 0; q < N; ++q) stmp.p[q] /= s->rho_ptime.p[q];
    }
    orc_abec_level L;
    ns_scalar_level(s, &L, comp, 0.0, -1.0, NULL);
    orc_fab bcval = orc_alloc(g->n, ORC_CELL, 1, 1);
    orc_copy_all(&bcval, &stmp);
    orc_fab cfb = orc_alloc(g->n, ORC_CELL, 1, 3);
    if (s->level > 0) {                         /* mlabec.setCoarseFineBC(&crsedata, ratio), Diffusion.cpp:1600-1609 */
        L.nbox = s->nbox; L.boxes = s->boxes;
        for (int d = 0; d < 3; ++d) L.cf_loc[d] = 0.5 * s->ratio * g->dx[d];
        orc_fab cd = crse_scalar_at(s, time_of(s, Sdata), comp, over_rho);
        orc_cf_interp_bndry(&L, s->ratio, &cd, &cfb);
        orc_free(&cd);
        orc_cf_set_bcval(&cfb, 1, 2);
    }
    orc_abec_applybc(&L, &stmp, slobc, shibc, 2, 1, &bcval);
    orc_fab tmp = orc_alloc(g->n, ORC_CELL, 0, 1);
    orc_abec_apply(&L, &tmp, &stmp);
    orc_cf_set_bcval(NULL, 0, 2);
    orc_free(&cfb);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(visc, i, j, k, 0) = A4(&tmp, i, j, k, 0);
    orc_fill_periodic(visc, g, ORC_CELL);
    first_order_extrap_cf(s, visc);
    first_order_extrap(visc, g);
    orc_free(&tmp); orc_free(&bcval); orc_free(&stmp);
    for (int d = 0; d < 3; ++d) orc_free(&L.b[d]);
}

/* NavierStokes::getViscTerms for velocity: visc = div tau(U(time)), then FillBoundary (+ FirstOrderExtrap at walls) */
void ns_get_visc_terms_vel(const orc_ns_state* s, orc_fab* visc /*3 comps, 1 ghost*/, const orc_fab* Sdata)
{
    const orc_geom* g = &s->g;
    orc_setval(visc, 1.e40);
    if (!is_diffusive_vel(s)) { orc_setval(visc, 0.0); return; }
    orc_fab stmp = fillpatch(s, Sdata, Xvel, 3, 1, s->bc_vel);
    orc_fab eta[3]; orc_fab* ep[3];
    make_eta(s, eta);
    for (int d = 0; d < 3; ++d) ep[d] = &eta[d];
    orc_fab tmp = orc_alloc(g->n, ORC_CELL, 0, 3);
    if (s->level > 0) {                         /* tensorop.setCoarseFineBC(&crsedata, ratio), Diffusion.cpp:1725-1736 */
        orc_fab cd = crse_vel_at(s, time_of(s, Sdata));
        orc_tensor_apply_cf(g, s->nbox, s->boxes, s->ratio, &tmp, &stmp, 0.0, -1.0, NULL, ep, s->vlobc, s->vhibc, 2, &cd);
        orc_free(&cd);
    } else
    orc_tensor_apply_bcn(g, &tmp, &stmp, 0.0, -1.0, NULL, ep, s->vlobc, s->vhibc, 2);
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(visc, i, j, k, n) = A4(&tmp, i, j, k, n);
    orc_fill_periodic(visc, g, ORC_CELL);
    first_order_extrap_cf(s, visc);
    first_order_extrap(visc, g);
    orc_free(&tmp); orc_free(&stmp);
    for (int d = 0; d < 3; ++d) orc_free(&eta[d]);
}

/* getForce: rho-weighted buoyancy in the last direction (Source/NS_getForce.cpp:117-137) */
static double force_vel(const orc_ns_state* s, int n, double rho)
{
    if (fabs(s->p.gravity) > 0.0001 && n == 2) return s->p.gravity * rho;
    return 0.0;
}

double ns_est_time_step(orc_ns_state* s)
{
    const orc_geom* g = &s->g;
    if (s->p.fixed_dt > 0.0) return s->p.fixed_dt;
    const double small = 1.0e-8;
    double estdt = 1.0e+20;
    const orc_fab* S = S_NEW(s);
    const orc_fab* Gp = GP_NEW(s);
    double umax[3] = {0, 0, 0}, fmax_[3] = {0, 0, 0};
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        if (s->cov.p && A4(&s->cov, i, j, k, 0) == 0.0) continue;
        double u = fabs(A4(S, i, j, k, n)); if (u > umax[n]) umax[n] = u;
        double rho = A4(S, i, j, k, Density);
        double rho_inv = 1.0 / rho;
        double f = force_vel(s, n, rho);
        f -= A4(Gp, i, j, k, n);
        f *= rho_inv;
        f = fabs(f); if (f > fmax_[n]) fmax_[n] = f;
    }
    for (int d = 0; d < 3; ++d) {
        if (umax[d] > small) estdt = fmin(estdt, g->dx[d] / umax[d]);
        if (fmax_[d] > small) estdt = fmin(estdt, sqrt(2.0 * g->dx[d] / fmax_[d]));
    }
    if (estdt < 1.0e+20) estdt *= s->p.cfl;
    else if (s->p.init_dt > 0.0) estdt = s->p.init_dt;          /* NavierStokesBase.cpp:1463-1481 */
    else { fprintf(stderr, "orc estTimeStep failed (zero velocity and force: set init_dt)\n"); abort(); }
    return estdt;
}

void ns_set_inflow_ghosts(const orc_ns_state* s, orc_fab* vel, double inflow_scale)
{
    const orc_geom* g = &s->g;
    /* inflow faces: the ghost cells hold the boundary value of the projected field (setPhysBoundaryValues before the scaling of
     * U_new, Projection.cpp:199-207): inflow velocity x inflow_scale (1/dt in level_project, 1 in the initial velocity projection,
     * 0 for the time-difference of a steady inflow in initialSyncProject) */
    for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
        if ((side == 0 ? s->nlobc[d] : s->nhibc[d]) != ORC_LO_INFLOW) continue;
        const double uin = (side == 0 ? s->ed_vel_lo[d * 3 + d] : s->ed_vel_hi[d * 3 + d]) * inflow_scale;
        int lo[3] = {-1, -1, -1}, hi[3] = {g->n[0], g->n[1], g->n[2]};
        lo[d] = hi[d] = side == 0 ? -1 : g->n[d];
        for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) A4(vel, i, j, k, d) = uin;
    }
}

/* Projection::doMLMGNodalProjection on ONE level (Projection.cpp:2385-2567).  sync != 0 (level_project only): the sync residuals of
 * Projection.cpp:367-377 are computed from the unprojected velocity and the solution and go into the sync registers (:401-431). */
static void nodal_project_level(orc_ns_state* s, orc_fab* vel /*3 comps 1 ghost, comps 0..2*/, orc_fab* phi, const orc_fab* sig,
                                int increment_gp, double inflow_scale, int sync, const orc_fab* rhcc /* cell source (-divu/dt ...) or NULL */)
{
    const orc_geom* g = &s->g;
    /* set_boundary_velocity + FillBoundary of vel ghost cells (periodic) */
    orc_fill_periodic(vel, g, ORC_CELL);
    ns_set_inflow_ghosts(s, vel, inflow_scale);
    const int want_crse = sync && s->fine != NULL;
    const int want_fine = sync && s->level > 0 && s->iteration == s->ncycle;
    orc_fab vold; vold.p = NULL;
    if (want_crse || want_fine) {
        vold = orc_alloc(g->n, ORC_CELL, 1, 3);
        for (int n = 0; n < 3; ++n)
        for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) A4(&vold, i, j, k, n) = A4(vel, i, j, k, n);
    }
    if (rhcc) orc_nodal_project_rhcc(g, vel, phi, sig, s->nlobc, s->nhibc, s->level == 0 ? NULL : &s->cov, rhcc, s->p.proj_tol, s->p.proj_abs_tol, &s->o, &s->st_nodal);
    else if (s->level == 0) orc_nodal_project(g, vel, phi, sig, s->nlobc, s->nhibc, s->p.proj_tol, s->p.proj_abs_tol, &s->o, &s->st_nodal);
    else orc_nodal_project_cov(g, vel, phi, sig, s->nlobc, s->nhibc, &s->cov, s->p.proj_tol, s->p.proj_abs_tol, &s->o, &s->st_nodal);
    /* Gp_new := grad(phi) or += (Projection.cpp:2549-2563), then FillPatch(Gp) */
    orc_fab gp = orc_alloc(g->n, ORC_CELL, 0, 3);
    orc_nodal_compgrad(g, &gp, phi);
    orc_fab* G = GP_NEW(s);
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        if (s->cov.p && A4(&s->cov, i, j, k, 0) == 0.0) continue;
        if (increment_gp) A4(G, i, j, k, n) += A4(&gp, i, j, k, n);
        else A4(G, i, j, k, n) = A4(&gp, i, j, k, n);
    }
    ns_fill_gp(s, G, 0.5 * (s->pt_new[0] + s->pt_new[1]));   /* FillPatch(Gradp_Type) at its current time, Projection.cpp:2564-2565 */
    orc_free(&gp);
    if (want_crse) {            /* crse_sync_reg->CrseInit(sync_resid_crse, geom, 1.0) */
        orc_fab r = amr_sync_resid_crse(s, &vold, phi, sig, rhcc);
        syncreg_crse_init(s->fine, &r, 1.0);
        orc_free(&r);
        orc_ndmf* rb = orc_sync_resid_crse_boxes(s, &vold, phi, sig, rhcc);          /* the same, box by box, into the literal register */
        orc_syncreg_crse_init(s->fine->sync_lit, rb, g, 1.0);
        orc_ndmf_destroy(rb);
    }
    if (want_fine) {            /* fine_sync_reg->FineAdd(sync_resid_fine, crse_geom, 1/crse_dt_ratio) */
        orc_fab r = amr_sync_resid_fine(s, &vold, phi, sig, rhcc);
        syncreg_fine_add(s, &r, 1.0 / (double)s->ncycle);
        orc_free(&r);
        orc_ndmf* rb = orc_sync_resid_fine_boxes(s, &vold, phi, sig, rhcc);
        orc_syncreg_fine_add(s->sync_lit, rb, &s->crse->g, 1.0 / (double)s->ncycle);
        orc_ndmf_destroy(rb);
    }
    if (vold.p) orc_free(&vold);
}

/* wrap the velocity comps of a state fab as a 3-comp view (same memory) */
static orc_fab vel_view(orc_fab* S) { orc_fab v = *S; v.nc = 3; return v; }

static void initial_velocity_project(orc_ns_state* s)
{
    const orc_geom* g = &s->g;
    if (s->p.init_vel_iter <= 0) { orc_setval(P_OLD(s), 0.0); orc_setval(GP_OLD(s), 0.0); return; }
    for (int iter = 0; iter < s->p.init_vel_iter; ++iter) {
        orc_fab* phi = P_OLD(s);
        orc_setval(phi, 0.0);
        orc_fab sig = orc_alloc(g->n, ORC_CELL, 1, 1);
        orc_setval(&sig, 1.0);       /* constant-density initial projection; scaleVar inverts: 1/1 */
        orc_fab v = vel_view(S_NEW(s));
        orc_fab rhcc; rhcc.p = NULL;
        if (s->have_divu) {                 /* rhcc = -getDivCond(cur_divu_time), Projection.cpp:732-743, 783-788 */
            rhcc = orc_alloc(g->n, ORC_CELL, 0, 1);
            for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&rhcc, i, j, k, 0) = -A4(S_NEW(s), i, j, k, s->Divu);
        }
        nodal_project_level(s, &v, phi, &sig, 0, 1.0, 0, rhcc.p ? &rhcc : NULL);
        if (rhcc.p) orc_free(&rhcc);
        orc_free(&sig);
        orc_setval(P_OLD(s), 0.0); orc_setval(P_NEW(s), 0.0);
        orc_setval(GP_OLD(s), 0.0); orc_setval(GP_NEW(s), 0.0);
    }
}

/* NavierStokesBase::advance_setup (NavierStokesBase.cpp:613-741) */
static void advance_setup(orc_ns_state* s, double dt, int iteration, int ncycle)
{
    const orc_geom* g = &s->g;
    s->iteration = iteration; s->ncycle = ncycle;
    if (s->fine) {
        orc_setval(&s->Vsync, 0.0); orc_setval(&s->Ssync, 0.0);          /* :643-650 */
        reg_setval(s->fine->reg_adv, 0.0); reg_setval(s->fine->reg_visc, 0.0);   /* :655-659 */
    }
    if (!s->initial_step && s->level > 0 && iteration == 1) {           /* initRhoAvg(0.5/ncycle), :685-687 (before the swap) */
        const double alpha = 0.5 / (double)ncycle;
        orc_setval(&s->rho_avg, 1.e200);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
            A4(&s->rho_avg, i, j, k, 0) = A4(S_NEW(s), i, j, k, Density) * alpha;
    }
    s->inew = 1 - s->inew;     /* swapTimeLevels: old <- new, new <- old storage */
    s->pnew = 1 - s->pnew;
    swap_time_levels(s, dt);
    /* make_rho_prev_time */
    orc_fab r = fillpatch(s, S_OLD(s), Density, 1, 1, &s->bc_scal[0]);
    orc_copy_all(&s->rho_ptime, &r);
    orc_free(&r);
}

static double predict_velocity(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_godunov_set_ppm(s->p.use_ppm);
    orc_fab Umf = fillpatch(s, S_OLD(s), Xvel, 3, 3, s->bc_vel);
    floor_small(&Umf);
    double cflmax = 0.0;
    for (int n = 0; n < 3; ++n) {
        double um = 0.0;
        for (int k = Umf.lo[2]; k <= Umf.hi[2]; ++k) for (int j = Umf.lo[1]; j <= Umf.hi[1]; ++j) for (int i = Umf.lo[0]; i <= Umf.hi[0]; ++i) {
            if (s->level > 0 && !ns_in_grown(s, i, j, k, 3)) continue;     /* norm0 over the FillPatch'ed boxes of the level, 3 ghost cells */
            double v = fabs(A4(&Umf, i, j, k, n)); if (v > um) um = v;
        }
        double c = dt * um / g->dx[n];
        if (n == 0 || c > cflmax) cflmax = c;
    }
    /* NavierStokesBase.cpp:4417-4422: on a refined level the ghost cells of the old Gradp are re-filled, the coarse data have changed */
    if (s->level > 0) ns_fill_gp(s, GP_OLD(s), 0.5 * (s->pt_old[0] + s->pt_old[1]));
    double tempdt = cflmax == 0 ? s->p.change_max : fmin(s->p.change_max, s->p.cfl / cflmax);
    orc_fab visc = orc_alloc(g->n, ORC_CELL, 1, 3);
    if (s->p.be_cn_theta != 1.0) ns_get_visc_terms_vel(s, &visc, S_OLD(s)); else orc_setval(&visc, 0.0);
    orc_fab Smf = fillpatch(s, S_OLD(s), Density, s->nscal, 3, s->bc_scal);
    orc_fab tf = orc_alloc(g->n, ORC_CELL, 1, 3);
    const orc_fab* Gp = GP_OLD(s);
    for (int n = 0; n < 3; ++n)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
        double rho = A4(&Smf, i, j, k, 0);
        A4(&tf, i, j, k, n) = (force_vel(s, n, rho) + A4(&visc, i, j, k, n) - A4(Gp, i, j, k, n)) / rho;
    }
    orc_fab* um[3] = {&s->umac[0], &s->umac[1], &s->umac[2]};
    orc_extrap_vel_to_faces(g, &Umf, &tf, um, dt, s->bc_vel, s->p.use_forces_in_trans);
    orc_free(&Umf); orc_free(&visc); orc_free(&Smf); orc_free(&tf);
    return dt * tempdt;
}

/* NavierStokesBase::create_umac_grown on a refined level (NavierStokesBase.cpp:1108-1311): ghost faces by FaceLinear interpolation of
 * the coarse mac velocities where no fine face exists, then the divergence fix of the outer face of every ghost cell that is not
 * covered and has exactly one face neighbour inside the level.  Whole-domain form: assumes that two boxes of a level are never
 * separated by a gap of exactly two cells (blocking factor >= 4), so that no face is the outer face of two ghost cells. */
static void create_umac_grown_fine(orc_ns_state* s, const orc_fab* divu /* 1 ghost, or NULL: 0 */)
{
    const orc_geom* g = &s->g;
    const int r = s->ratio;
    for (int d = 0; d < 3; ++d) {
        orc_fab* f = &s->umac[d];
        const orc_fab* uc = &s->crse->umac[d];
        for (int k = f->lo[2]; k <= f->hi[2]; ++k) for (int j = f->lo[1]; j <= f->hi[1]; ++j) for (int i = f->lo[0]; i <= f->hi[0]; ++i) {
            int m[3] = {i, j, k}; m[d] -= 1;
            if (ns_covered(s, i, j, k) || ns_covered(s, m[0], m[1], m[2])) continue;     /* a face of the level */
            const int fi[3] = {i, j, k};
            int c[3];
            for (int e = 0; e < 3; ++e) c[e] = fi[e] >= 0 ? fi[e] / r : -((-fi[e] + r - 1) / r);
            int ok = 1;
            for (int e = 0; e < 3; ++e) if (c[e] < uc->lo[e] || c[e] + (e == d ? 1 : 0) > uc->hi[e]) ok = 0;
            if (!ok) continue;                                                           /* beyond the coarse ghost faces: never used */
            const int rem = fi[d] - c[d] * r;
            double v;
            if (rem == 0) v = A4(uc, c[0], c[1], c[2], 0);
            else {
                const double w = (double)rem / (double)r;
                int cp[3] = {c[0], c[1], c[2]}; cp[d] += 1;
                v = (1.0 - w) * A4(uc, c[0], c[1], c[2], 0) + w * A4(uc, cp[0], cp[1], cp[2], 0);
            }
            A4(f, i, j, k, 0) = v;
        }
        orc_fill_periodic(f, g, ORC_FACE[d]);
    }
    orc_fab *u = &s->umac[0], *v = &s->umac[1], *w = &s->umac[2];
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
        const int idx[3] = {i, j, k};
        int outside = 0;
        for (int e = 0; e < 3; ++e) if (!g->periodic[e] && (idx[e] < 0 || idx[e] > g->n[e] - 1)) outside = 1;
        if (outside || ns_covered(s, i, j, k)) continue;
        int count = 0;
        for (int e = 0; e < 3; ++e) for (int sg = -1; sg <= 1; sg += 2) { int q[3] = {i, j, k}; q[e] += sg; count += ns_covered(s, q[0], q[1], q[2]); }
        if (count != 1) continue;
        const double dux = (A4(u, i + 1, j, k, 0) - A4(u, i, j, k, 0)) / g->dx[0];
        const double duy = (A4(v, i, j + 1, k, 0) - A4(v, i, j, k, 0)) / g->dx[1];
        const double duz = (A4(w, i, j, k + 1, 0) - A4(w, i, j, k, 0)) / g->dx[2];
        const double dv = divu ? A4(divu, i, j, k, 0) : 0.0;                /* the divergence constraint of the cell, NavierStokesBase.cpp:1235 */
        if (ns_covered(s, i + 1, j, k)) A4(u, i, j, k, 0) = A4(u, i + 1, j, k, 0) + g->dx[0] * (duy + duz - dv);
        else if (ns_covered(s, i - 1, j, k)) A4(u, i + 1, j, k, 0) = A4(u, i, j, k, 0) - g->dx[0] * (duy + duz - dv);
        if (ns_covered(s, i, j + 1, k)) A4(v, i, j, k, 0) = A4(v, i, j + 1, k, 0) + g->dx[1] * (dux + duz - dv);
        else if (ns_covered(s, i, j - 1, k)) A4(v, i, j + 1, k, 0) = A4(v, i, j, k, 0) - g->dx[1] * (dux + duz - dv);
        if (ns_covered(s, i, j, k + 1)) A4(w, i, j, k, 0) = A4(w, i, j, k + 1, 0) + g->dx[2] * (dux + duy - dv);
        else if (ns_covered(s, i, j, k - 1)) A4(w, i, j, k + 1, 0) = A4(w, i, j, k, 0) - g->dx[2] * (dux + duy - dv);
    }
    for (int d = 0; d < 3; ++d) orc_fill_periodic(&s->umac[d], g, ORC_FACE[d]);
}

static orc_fab dbg_umac[3];
orc_fab* orc_dbg_umac(int d) { return &dbg_umac[d]; }
/* NavierStokes::calc_divu (NavierStokes.cpp:1876-1958): divu = div(lambda grad T) / (rho T) at the new (use_new) or old time, valid cells */
void ns_calc_divu(orc_ns_state* s, int use_new)
{
    const orc_geom* g = &s->g;
    if (!s->have_divu) return;
    orc_fab* S = use_new ? S_NEW(s) : S_OLD(s);
    if (!(s->scal_diff[s->Temp - Density] > 0.0)) {
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(S, i, j, k, s->Divu) = 0.0;
        return;
    }
    orc_fab visc = orc_alloc(g->n, ORC_CELL, 1, 1);
    ns_get_visc_terms_scalar(s, &visc, S, s->Temp);
    const orc_fab* rho = use_new ? &s->rho_ctime : &s->rho_ptime;          /* get_rho(time) */
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(S, i, j, k, s->Divu) = A4(&visc, i, j, k, 0) / (A4(rho, i, j, k, 0) * A4(S, i, j, k, s->Temp));
    orc_free(&visc);
}
/* NavierStokesBase::calc_dsdt (NavierStokesBase.cpp:818-858): dsdt_new = (divu_new - divu_old) / dt */
void ns_calc_dsdt(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    if (!s->have_divu) return;
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(S_NEW(s), i, j, k, s->Dsdt) = (A4(S_NEW(s), i, j, k, s->Divu) - A4(S_OLD(s), i, j, k, s->Divu)) / dt;
}
/* getDivCond(ng, prev_time) (+ dt/2 getDsdt(ng, prev_time)): the constraint of the MAC projection (create_mac_rhs, NavierStokesBase.cpp:
 * 1038-1065) and of the advective update (NavierStokesBase.cpp:3377-3424, NavierStokes.cpp:712-733); zero without divu */
orc_fab ns_divu_half(const orc_ns_state* s, double dt, int ng, int with_dsdt)
{
    const orc_geom* g = &s->g;
    if (!s->have_divu) return orc_alloc(g->n, ORC_CELL, ng, 1);
    orc_fab d = ns_fillpatch_time(s, s->st_old, 0, s->Divu, 1, ng);
    if (with_dsdt) {
        orc_fab e = ns_fillpatch_time(s, s->st_old, 0, s->Dsdt, 1, ng);
        const size_t N = orc_npts(&d);
        for (size_t q = 0; q < N; ++q) d.p[q] += 0.5 * dt * e.p[q];
        orc_free(&e);
    }
    return d;
}

static void mac_project(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_fab* phi = &s->mac_phi;                                 /* mac_phi_crse[level]: kept as the coarse/fine data of the next finer level */
    orc_setval(phi, 0.0);
    /* MacProj.cpp:262-263: S_old density ghost cells overwritten with rho(time) incl. 1 ghost */
    orc_fab* So = S_OLD(s);
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i)
        A4(So, i, j, k, Density) = A4(&s->rho_ptime, i, j, k, 0);
    orc_fab* um[3] = {&s->umac[0], &s->umac[1], &s->umac[2]};
    orc_mg_opts o = s->o; o.maxorder = 4;
    if (getenv("ORC_DBG_UMAC") && s->level == atoi(getenv("ORC_DBG_UMAC"))) {
        for (int d = 0; d < 3; ++d) { if (dbg_umac[d].p) orc_free(&dbg_umac[d]); dbg_umac[d] = orc_alloc(g->n, ORC_FACE[d], 1, 1); orc_copy_all(&dbg_umac[d], &s->umac[d]); }
    }
    orc_fab mac_rhs = ns_divu_half(s, dt, 1, 1);                 /* create_mac_rhs(mac_rhs, 1, time, dt), NavierStokes.cpp:592-596 */
    const orc_fab* Sp = s->have_divu ? &mac_rhs : NULL;
    if (s->level == 0) {
        orc_mac_project(g, um, &s->rho_ptime, Sp, phi, 2.0 / dt, s->lobc, s->hibc, s->p.mac_tol, s->p.mac_abs_tol, &o, &s->st_mac);
        /* create_umac_grown at level 0: FillPatchSingleLevel (periodic ghost faces) */
        for (int d = 0; d < 3; ++d) fill_ghosts(s, &s->umac[d], ORC_FACE[d]);
    } else {
        orc_fill_periodic(&s->crse->mac_phi, &s->crse->g, ORC_CELL);
        orc_mac_project_cf(g, um, &s->rho_ptime, Sp, phi, 2.0 / dt, s->lobc, s->hibc, s->nbox, s->boxes, s->ratio, &s->crse->mac_phi,
                           s->p.mac_tol, s->p.mac_abs_tol, &o, &s->st_mac);
    }
    orc_fill_periodic(phi, g, ORC_CELL);
    /* MAC registers (MacProj.cpp:304-348): fluxes = u_mac * area */
    for (int d = 0; d < 3; ++d) {
        const double area = g->dx[(d + 1) % 3] * g->dx[(d + 2) % 3];
        if (s->fine) reg_crse_init(s->fine, s->fine->reg_mac, &s->umac[d], d, 0, 0, 1, -1.0 * area, 0);
        if (s->level > 0) reg_fine_add(s, s->reg_mac, &s->umac[d], d, 0, 0, 1, area / (double)s->ncycle);
    }
    if (s->level > 0) create_umac_grown_fine(s, Sp);
    orc_free(&mac_rhs);
    /* "BDS needs physical BCs filled" (NavierStokesBase.cpp:1097-1105): ghost faces outside a non-periodic domain face = the nearest
     * face inside or on the boundary (first-order extrapolation; the boundary functor itself is upstream) */
    if (s->p.use_ppm == 2)
        for (int d = 0; d < 3; ++d) {
            orc_fab* u = &s->umac[d];
            for (int k = u->lo[2]; k <= u->hi[2]; ++k) for (int j = u->lo[1]; j <= u->hi[1]; ++j) for (int i = u->lo[0]; i <= u->hi[0]; ++i) {
                int q[3] = {i, j, k}, out = 0;
                for (int e = 0; e < 3; ++e) {
                    if (g->periodic[e]) continue;
                    const int hi = g->n[e] - 1 + (e == d ? 1 : 0);
                    if (q[e] < 0) { q[e] = 0; out = 1; } else if (q[e] > hi) { q[e] = hi; out = 1; }
                }
                if (out) A4(u, i, j, k, 0) = A4(u, q[0], q[1], q[2], 0);
            }
        }
}

/* NavierStokesBase::ComputeAofs, flux-register part (NavierStokesBase.cpp:5075-5096): CrseAdd into the register of the next finer
 * level, FineAdd into the level's own; YAFluxRegister semantics written as CrseInit(-dt) / FineAdd(+dt), see orc_amr.c */
static void adv_registers(orc_ns_state* s, orc_fab* flux[3], int state_indx, int ncomp, double dt)
{
    for (int d = 0; d < 3; ++d) {
        if (s->fine) reg_crse_init(s->fine, s->fine->reg_adv, flux[d], d, 0, state_indx, ncomp, -dt, 1);
        if (s->level > 0) reg_fine_add(s, s->reg_adv, flux[d], d, 0, state_indx, ncomp, dt);
    }
}

static void velocity_advection(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_godunov_set_ppm(s->p.use_ppm);
    const int mom = s->p.do_mom_diff;
    orc_fab Umf = fillpatch(s, S_OLD(s), Xvel, 3, 3, s->bc_vel);
    if (mom) {
        /* NavierStokesBase.cpp:3397-3413: the advected state is the momentum rho^n u^n, ghost cells included (both factors
         * FillPatched with their own physical BC) */
        orc_fab Rmf = fillpatch(s, S_OLD(s), Density, 1, 3, &s->bc_scal[0]);
        const size_t N = orc_npts(&Umf);
        for (int n = 0; n < 3; ++n) for (size_t q = 0; q < N; ++q) Umf.p[q + N * n] *= Rmf.p[q];
        orc_free(&Rmf);
    }
    orc_fab Smf = fillpatch(s, S_OLD(s), Density, s->nscal, 1, s->bc_scal);
    orc_fab visc = orc_alloc(g->n, ORC_CELL, 1, 3);
    if (s->p.be_cn_theta != 1.0) ns_get_visc_terms_vel(s, &visc, S_OLD(s)); else orc_setval(&visc, 0.0);
    orc_fab tf = orc_alloc(g->n, ORC_CELL, 1, 3);
    orc_fab divu = ns_divu_half(s, dt, 1, 1);              /* getDivCond(prev_time) + dt/2 getDsdt(prev_time): NavierStokesBase.cpp:3377-3424, NavierStokes.cpp:712-733 */
    const orc_fab* Gp = GP_OLD(s);
    for (int n = 0; n < 3; ++n)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
        double rho = A4(&Smf, i, j, k, 0);
        double t = force_vel(s, n, rho) + A4(&visc, i, j, k, n) - A4(Gp, i, j, k, n);
        if (!mom) t /= rho;                     /* NavierStokesBase.cpp:3459-3466: convective form only */
        A4(&tf, i, j, k, n) = t;
    }
    int iconserv[3] = {mom, mom, mom};          /* NS_setup.cpp:297-301: velocity advectionType = Conservative */
    orc_fab* um[3] = {&s->umac[0], &s->umac[1], &s->umac[2]};
    orc_fab fl[3]; orc_fab* flp[3];
    for (int d = 0; d < 3; ++d) { fl[d] = orc_alloc(g->n, ORC_FACE[d], 0, 3); flp[d] = &fl[d]; }
    orc_compute_aofs(g, &s->aofs, Xvel, &Umf, 3, &tf, &divu, um, iconserv, dt, s->bc_vel, 1, s->p.use_forces_in_trans, NULL, flp);
    adv_registers(s, flp, Xvel, 3, dt);
    for (int d = 0; d < 3; ++d) orc_free(&fl[d]);
    orc_free(&Umf); orc_free(&Smf); orc_free(&visc); orc_free(&tf); orc_free(&divu);
}

static void scalar_advection(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_godunov_set_ppm(s->p.use_ppm);
    orc_fab Smf = fillpatch(s, S_OLD(s), Density, s->nscal, 3, s->bc_scal);
    floor_small(&Smf);
    orc_fab tf = orc_alloc(g->n, ORC_CELL, 1, s->nscal);   /* getForce = 0, visc = 0 (non-diffusive scalars) */
    orc_fab divu = ns_divu_half(s, dt, 1, 1);              /* getDivCond(prev_time) + dt/2 getDsdt(prev_time): NavierStokesBase.cpp:3377-3424, NavierStokes.cpp:712-733 */
    int iconserv[ORC_MAXSCAL];                          /* advectionType, NS_setup.cpp:297-320 */
    for (int n = 0; n < s->nscal; ++n) iconserv[n] = s->scal_cons[n];
    orc_fab visc = orc_alloc(g->n, ORC_CELL, 1, 1);
    for (int n = 1; n < s->nscal; ++n) {                /* n = 0: density, tf += visc = 0 */
        if (s->p.be_cn_theta != 1.0) ns_get_visc_terms_scalar(s, &visc, S_OLD(s), Density + n); else orc_setval(&visc, 0.0);
        for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
            const double rho = A4(&Smf, i, j, k, 0);
            if (Density + n == s->Temp) A4(&tf, i, j, k, n) = (A4(&tf, i, j, k, n) + A4(&visc, i, j, k, 0)) / rho;   /* NavierStokes.cpp:766-778 */
            else if (s->scal_cons[n]) A4(&tf, i, j, k, n) += A4(&visc, i, j, k, 0);                                   /* :780-792 */
            else A4(&tf, i, j, k, n) = A4(&tf, i, j, k, n) / rho + A4(&visc, i, j, k, 0);                              /* :794-806: convective, tf/rho + visc */
        }
    }
    orc_free(&visc);
    orc_fab* um[3] = {&s->umac[0], &s->umac[1], &s->umac[2]};
    orc_fab fl[3]; orc_fab* flp[3];
    for (int d = 0; d < 3; ++d) { fl[d] = orc_alloc(g->n, ORC_FACE[d], 0, s->nscal); flp[d] = &fl[d]; }
    orc_compute_aofs(g, &s->aofs, Density, &Smf, s->nscal, &tf, &divu, um, iconserv, dt, s->bc_scal, 0, s->p.use_forces_in_trans, NULL, flp);
    adv_registers(s, flp, Density, s->nscal, dt);
    for (int d = 0; d < 3; ++d) orc_free(&fl[d]);
    orc_free(&Smf); orc_free(&tf); orc_free(&divu);
}

/* NavierStokesBase::ConservativeScalMinMax / ConvectiveScalMinMax (NavierStokesBase.cpp:4256-4368): the new value (per unit mass if
 * conservative) clipped to the min / max of the FillPatch'ed old data over the 27 neighbours.  As written upstream the running maximum
 * starts from std::numeric_limits<Real>::min() -- the smallest POSITIVE double, not the lowest. */
static void scal_min_max(orc_ns_state* s, int comp, int conservative)
{
    const orc_geom* g = &s->g;
    orc_fab* Sn = S_NEW(s);
    orc_fab So = fillpatch(s, S_OLD(s), Density, s->nscal, 1, s->bc_scal);       /* FillPatchIterator(S_old, 1, prev_time, Density, num_scalars) */
    const int oc = comp - Density;
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        if (s->cov.p && A4(&s->cov, i, j, k, 0) == 0.0) continue;
        double smn = 1.7976931348623157e308, smx = 2.2250738585072014e-308;
        for (int kk = -1; kk <= 1; ++kk) for (int jj = -1; jj <= 1; ++jj) for (int ii = -1; ii <= 1; ++ii) {
            const double v = conservative ? A4(&So, i + ii, j + jj, k + kk, oc) / A4(&So, i + ii, j + jj, k + kk, 0) : A4(&So, i + ii, j + jj, k + kk, oc);
            smn = fmin(smn, v); smx = fmax(smx, v);
        }
        if (conservative) { const double rn = A4(Sn, i, j, k, Density); A4(Sn, i, j, k, comp) = fmin(fmax(A4(Sn, i, j, k, comp) / rn, smn), smx) * rn; }
        else A4(Sn, i, j, k, comp) = fmin(fmax(A4(Sn, i, j, k, comp), smn), smx);
    }
    orc_free(&So);
}

static void scalar_update_rho(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_fab *Sn = S_NEW(s), *So = S_OLD(s);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(Sn, i, j, k, Density) = A4(So, i, j, k, Density) - dt * A4(&s->aofs, i, j, k, Density);
    if (s->p.do_denminmax) scal_min_max(s, Density, 1);                                 /* :2771-2788 */
    /* make_rho_curr_time + get_rho_half_time */
    orc_fab r = fillpatch(s, Sn, Density, 1, 1, &s->bc_scal[0]);
    orc_copy_all(&s->rho_ctime, &r);
    orc_free(&r);
    size_t N = orc_npts(&s->rho_half);
    for (size_t q = 0; q < N; ++q) s->rho_half.p[q] = 0.5 * (s->rho_ptime.p[q] + s->rho_ctime.p[q]);
}

static void scalar_update_tracers(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_fab *Sn = S_NEW(s), *So = S_OLD(s);
    for (int sigma = Tracer; sigma < s->nstate; ++sigma) {
        const int cons = s->scal_cons[sigma - Density];
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
            double rho = A4(So, i, j, k, Density) - 0.5 * dt * A4(&s->aofs, i, j, k, Density);
            double tf = 0.0;
            if (cons) A4(Sn, i, j, k, sigma) = A4(So, i, j, k, sigma) + dt * (-A4(&s->aofs, i, j, k, sigma) + tf);   /* NavierStokesBase.cpp:2889-2891 */
            else
            A4(Sn, i, j, k, sigma) = A4(So, i, j, k, sigma) + dt * (-A4(&s->aofs, i, j, k, sigma) + tf / rho);
        }
        if (s->p.do_scalminmax) scal_min_max(s, sigma, cons);                           /* :2907-2935 */
    }
}

/* NavierStokes::scalar_diffusion_update -> Diffusion::diffuse_scalar for the tracer (rho_flag 0, Laplacian_S;
 * reference Source/NavierStokes.cpp:867-1000, Source/Diffusion.cpp:207-599): Crank-Nicolson
 *   (1 - theta dt div beta grad) S_new = S* + (1-theta) dt div beta grad S_old */
static void scalar_diffusion_update_one(orc_ns_state* s, double dt, int sigma)
{
    const orc_geom* g = &s->g;
    if (!is_diffusive_scal(s, sigma)) return;
    const double theta = s->p.be_cn_theta;
    const int sn = sigma - Density, rho_flag = s->scal_rho_flag[sn];
    const int cons = rho_flag == 2;         /* diffusionType Laplacian_SoverRho -> rho_flag 2 (NS_setup.cpp:308, Diffusion.cpp:1870-1873) */
    const int* slobc = s->slobc + 3 * sn; const int* shibc = s->shibc + 3 * sn;
    orc_fab *Sn = S_NEW(s), *So = S_OLD(s);
    orc_fab Rhs = orc_alloc(g->n, ORC_CELL, 0, 1);
    const int want_flux = s->fine != NULL || s->level > 0;        /* viscous flux registers, NavierStokes.cpp:949-990 */
    orc_fab fl[3]; orc_fab* flp[3];
    for (int d = 0; d < 3; ++d) { fl[d] = orc_alloc(g->n, ORC_FACE[d], 0, 1); orc_setval(&fl[d], 0.0); flp[d] = &fl[d]; }
    orc_fab cfb = orc_alloc(g->n, ORC_CELL, 1, 3);
    if (theta != 1.0) {
        /* FillPatch(S_old, ng 1) then opn.setLevelBC(Soln = S_old tracer with ghosts); a = 0, b = -(1-theta) dt */
        orc_fab Soln = fillpatch(s, So, sigma, 1, 1, &s->bc_scal[sn]);
        if (cons) {             /* Diffusion.cpp:396-413: Soln = S_old / rho_old on the grown box */
            orc_fab R = fillpatch(s, So, Density, 1, 1, &s->bc_scal[0]);
            const size_t N = orc_npts(&Soln);
            for (size_t q = 0; q < N; ++q) Soln.p[q] /= R.p[q];
            orc_free(&R);
        }
        orc_fab bcval = orc_alloc(g->n, ORC_CELL, 1, 1);
        orc_copy_all(&bcval, &Soln);
        orc_abec_level Ln;
        ns_scalar_level(s, &Ln, sigma, 0.0, -(1.0 - theta) * dt, NULL);
        if (s->level > 0) {                     /* opn.setCoarseFineBC(Solnc = coarse S_old (/ rho_old), ratio), Diffusion.cpp:376-396 */
            Ln.nbox = s->nbox; Ln.boxes = s->boxes;
            for (int d = 0; d < 3; ++d) Ln.cf_loc[d] = 0.5 * s->ratio * g->dx[d];
            orc_fab cd = crse_scalar_at(s, s->st_old, sigma, cons);
            orc_cf_interp_bndry(&Ln, s->ratio, &cd, &cfb);
            orc_free(&cd);
            orc_cf_set_bcval(&cfb, 1, 2);
        }
        orc_abec_applybc(&Ln, &Soln, slobc, shibc, 2, 1, &bcval);
        orc_abec_apply(&Ln, &Rhs, &Soln);
        if (want_flux) orc_abec_extensive_flux(&Ln, flp, &Soln, 1.0 - theta, 0);     /* fluxn: computeExtensiveFluxes(..., -b/dt), Diffusion.cpp:437-438 */
        orc_cf_set_bcval(NULL, 0, 2);
        for (int d = 0; d < 3; ++d) orc_free(&Ln.b[d]);
        orc_free(&Soln); orc_free(&bcval);
    }
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(&Rhs, i, j, k, 0) += A4(Sn, i, j, k, sigma) * (rho_flag == 1 ? A4(&s->rho_half, i, j, k, 0) : 1.0);   /* Diffusion.cpp:476-486 */
    double m = 0.0;     /* get_scaled_abs_tol: visc_tol * ||Rhs||inf (one component) */
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        if (s->level > 0 && A4(&s->cov, i, j, k, 0) == 0.0) continue;
        double v = fabs(A4(&Rhs, i, j, k, 0)); if (v > m) m = v;
    }
    const double tol_abs = s->p.visc_tol * m;
    orc_fab Soln = fillpatch(s, Sn, sigma, 1, 1, &s->bc_scal[sn]);     /* FillPatch(S_new, ng 1): initial guess + level BC */
    orc_fab acoef = orc_alloc(g->n, ORC_CELL, 0, 1);
    orc_setval(&acoef, 1.0);                                            /* computeAlpha, rho_flag 0: alpha = 1 */
    if (cons) {                 /* rho_flag 2: Soln = S_new / rho_new on the grown box (Diffusion.cpp:520-540), alpha = rho_new (:1380-1383) */
        orc_fab R = fillpatch(s, Sn, Density, 1, 1, &s->bc_scal[0]);
        const size_t N = orc_npts(&Soln);
        for (size_t q = 0; q < N; ++q) Soln.p[q] /= R.p[q];
        orc_free(&R);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
            A4(&acoef, i, j, k, 0) = A4(Sn, i, j, k, Density);
    } else if (rho_flag == 1) { /* RhoInverse_Laplacian_S: alpha = rho_half (Diffusion.cpp:551-556, computeAlpha :1380-1383) */
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
            A4(&acoef, i, j, k, 0) = A4(&s->rho_half, i, j, k, 0);
    }
    orc_abec_level L;
    ns_scalar_level(s, &L, sigma, 1.0, theta * dt, &acoef);
    orc_mg_opts o = s->o; o.maxorder = 2;                                /* Diffusion::max_order = 2 */
    orc_mg_stats st;
    if (s->level > 0) {                         /* opnp1.setCoarseFineBC(coarse S_new (/ rho_new), ratio), Diffusion.cpp:506-518 */
        L.nbox = s->nbox; L.boxes = s->boxes;
        for (int d = 0; d < 3; ++d) L.cf_loc[d] = 0.5 * s->ratio * g->dx[d];
        orc_fab cd = crse_scalar_at(s, s->st_new, sigma, cons);
        orc_cf_interp_bndry(&L, s->ratio, &cd, &cfb);
        orc_free(&cd);
        orc_abec_solve_cf(&L, &Soln, &Rhs, slobc, shibc, &cfb, s->p.visc_tol, tol_abs, &o, &st);
    } else
    orc_abec_solve(&L, &Soln, &Rhs, slobc, shibc, s->p.visc_tol, tol_abs, &o, &st);
    s->st_scal = st;
    if (want_flux) {                            /* fluxnp1 = theta * area * (-D grad s_new), Diffusion.cpp:569-570; registers NavierStokes.cpp:949-990 */
        orc_cf_set_bcval(s->level > 0 ? &cfb : NULL, 1, 2);
        orc_abec_extensive_flux(&L, flp, &Soln, theta, 1);
        orc_cf_set_bcval(NULL, 0, 2);
        for (int d = 0; d < 3; ++d) {
            if (s->level > 0) reg_fine_add(s, s->reg_visc, &fl[d], d, 0, sigma, 1, dt);
            if (s->fine) reg_crse_init(s->fine, s->fine->reg_visc, &fl[d], d, 0, sigma, 1, -dt, 0);
        }
    }
    for (int d = 0; d < 3; ++d) orc_free(&fl[d]);
    orc_free(&cfb);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(Sn, i, j, k, sigma) = A4(&Soln, i, j, k, 0) * (cons ? A4(Sn, i, j, k, Density) : 1.0);    /* Diffusion.cpp:583-590 */
    for (int d = 0; d < 3; ++d) orc_free(&L.b[d]);
    orc_free(&Soln); orc_free(&acoef); orc_free(&Rhs);
}

static void scalar_diffusion_update(orc_ns_state* s, double dt)
{
    for (int sigma = Tracer; sigma < s->nstate; ++sigma) scalar_diffusion_update_one(s, dt, sigma);   /* NavierStokes.cpp:912-1000 */
}

static void velocity_advection_update(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_fab *Un = S_NEW(s), *Uo = S_OLD(s);
    const orc_fab* Gp = GP_OLD(s);
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        double scal_rho = 0.5 * (A4(Uo, i, j, k, Density) + A4(Un, i, j, k, Density));
        double force = force_vel(s, n, scal_rho);
        if (s->initial_iter && is_diffusive_vel(s)) force = 0.0;
        double rh = A4(&s->rho_half, i, j, k, 0);
        double velold = A4(Uo, i, j, k, n);
        if (s->p.do_mom_diff) {                 /* NavierStokesBase.cpp:3609-3616 */
            velold *= A4(Uo, i, j, k, Density);
            double v = velold - dt * A4(&s->aofs, i, j, k, n) + dt * force - dt * A4(Gp, i, j, k, n);
            A4(Un, i, j, k, n) = v / A4(Un, i, j, k, Density);
        } else
        A4(Un, i, j, k, n) = velold - dt * A4(&s->aofs, i, j, k, n) + dt * force / rh - dt * A4(Gp, i, j, k, n) / rh;
    }
}

static void initial_velocity_diffusion_update(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    if (!is_diffusive_vel(s)) return;
    orc_fab *Un = S_NEW(s), *Uo = S_OLD(s);
    const orc_fab* Gp = GP_OLD(s);
    orc_fab visc = orc_alloc(g->n, ORC_CELL, 1, 3);
    if (s->p.be_cn_theta != 1.0) ns_get_visc_terms_vel(s, &visc, Uo); else orc_setval(&visc, 0.0);
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        double force = force_vel(s, n, A4(Uo, i, j, k, Density));
        force += A4(&visc, i, j, k, n) - A4(Gp, i, j, k, n);
        if (!s->p.do_mom_diff) force /= A4(&s->rho_half, i, j, k, 0);
        force -= A4(&s->aofs, i, j, k, n);
        if (s->p.do_mom_diff)                   /* NavierStokesBase.cpp:3737-3739 */
            A4(Un, i, j, k, n) = (force * dt + A4(Uo, i, j, k, n) * A4(Uo, i, j, k, Density)) / A4(Un, i, j, k, Density);
        else
        A4(Un, i, j, k, n) = A4(Uo, i, j, k, n) + force * dt;
    }
    orc_free(&visc);
}

/* Diffusion::diffuse_tensor_velocity (rho_flag = 1, or 3 with do_mom_diff: NavierStokes.cpp:1016) */
static void velocity_diffusion_update(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    if (!is_diffusive_vel(s)) return;
    const int mom = s->p.do_mom_diff;
    const double theta = s->p.be_cn_theta;
    orc_fab *Un = S_NEW(s), *Uo = S_OLD(s);
    orc_fab eta[3]; orc_fab* ep[3];
    make_eta(s, eta);
    for (int d = 0; d < 3; ++d) ep[d] = &eta[d];
    orc_fab Rhs = orc_alloc(g->n, ORC_CELL, 0, 3);
    const int want_flux = s->fine != NULL || s->level > 0;        /* do_reflux && (level < finest_level || level > 0), Diffusion.cpp:790-796, 932-956 */
    orc_fab fl[3]; orc_fab* flp[3];
    for (int d = 0; d < 3; ++d) { fl[d] = orc_alloc(g->n, ORC_FACE[d], 0, 3); orc_setval(&fl[d], 0.0); flp[d] = &fl[d]; }
    if (theta != 1.0) {
        orc_fab Soln = fillpatch(s, Uo, Xvel, 3, 1, s->bc_vel);
        if (s->level > 0) {                     /* crsedata at prev_time, Diffusion.cpp:733-744 */
            orc_fab cd = crse_vel_at(s, s->st_old);
            orc_tensor_apply_cf(g, s->nbox, s->boxes, s->ratio, &Rhs, &Soln, 0.0, -(1.0 - theta) * dt, NULL, ep, s->vlobc, s->vhibc, 2, &cd);
            if (want_flux) orc_tensor_extensive_flux(g, s->nbox, s->boxes, s->ratio, flp, &Soln, ep, 1.0 - theta, 0, &cd, 2);
            orc_free(&cd);
        } else {
            orc_tensor_apply_bcn(g, &Rhs, &Soln, 0.0, -(1.0 - theta) * dt, NULL, ep, s->vlobc, s->vhibc, 2);
            if (want_flux) orc_tensor_extensive_flux(g, 0, NULL, 2, flp, &Soln, ep, 1.0 - theta, 0, NULL, 2);
        }
        orc_free(&Soln);
    }
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        /* Diffusion.cpp:819-825: state overwritten with rho u*; rho_flag 3 (do_mom_diff) multiplies by the OLD density */
        A4(Un, i, j, k, n) *= mom ? A4(Uo, i, j, k, Density) : A4(&s->rho_half, i, j, k, 0);
        A4(&Rhs, i, j, k, n) += A4(Un, i, j, k, n);
    }
    /* tol_abs = visc_tol * mean_n ||Rhs_n||inf (get_scaled_abs_tol) */
    double avg = 0.0;
    for (int n = 0; n < 3; ++n) {
        double m = 0.0;
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
            if (s->level > 0 && A4(&s->cov, i, j, k, 0) == 0.0) continue;
            double v = fabs(A4(&Rhs, i, j, k, n)); if (v > m) m = v;
        }
        avg += (1.0 / 3.0) * m;
    }
    const double tol_abs = s->p.visc_tol * avg;
    orc_fab Soln = fillpatch(s, Un, Xvel, 3, 1, s->bc_vel);   /* initial guess = FillPatch(U_new) = rho u* */
    orc_fab acoef = orc_alloc(g->n, ORC_CELL, 0, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(&acoef, i, j, k, 0) = mom ? A4(Un, i, j, k, Density) : A4(&s->rho_half, i, j, k, 0);   /* Diffusion.cpp:893: rho_flag 3 -> rho_new */
    orc_mg_opts o = s->o; o.maxorder = 2;
    if (s->level > 0) {                         /* crsedata at cur_time, Diffusion.cpp:876-887 */
        orc_fab cd = crse_vel_at(s, s->st_new);
        orc_tensor_solve_cf(g, s->nbox, s->boxes, s->ratio, &Soln, &Rhs, 1.0, theta * dt, &acoef, ep, s->vlobc, s->vhibc, &cd, s->p.visc_tol, tol_abs, &o, &s->st_visc);
        if (want_flux) orc_tensor_extensive_flux(g, s->nbox, s->boxes, s->ratio, flp, &Soln, ep, theta, 1, &cd, 2);
        orc_free(&cd);
    } else {
        orc_tensor_solve_bcn(g, &Soln, &Rhs, 1.0, theta * dt, &acoef, ep, s->vlobc, s->vhibc, s->p.visc_tol, tol_abs, &o, &s->st_visc);
        if (want_flux) orc_tensor_extensive_flux(g, 0, NULL, 2, flp, &Soln, ep, theta, 1, NULL, 2);
    }
    if (want_flux)
        for (int d = 0; d < 3; ++d) {
            if (s->level > 0) reg_fine_add(s, s->reg_visc, &fl[d], d, 0, Xvel, 3, dt);                          /* :946-949 */
            if (s->fine) reg_crse_init(s->fine, s->fine->reg_visc, &fl[d], d, 0, Xvel, 3, -dt, 0);              /* :950-954 */
        }
    for (int d = 0; d < 3; ++d) orc_free(&fl[d]);
    for (int n = 0; n < 3; ++n)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i)
        if (s->level == 0 || ns_in_grown(s, i, j, k, 1)) A4(Un, i, j, k, n) = A4(&Soln, i, j, k, n);
    orc_free(&Soln); orc_free(&acoef); orc_free(&Rhs);
    for (int d = 0; d < 3; ++d) orc_free(&eta[d]);
}

/* trilinear interpolation of a coarse nodal array at fine node (i,j,k) (amrex::NodeBilinear) */
static double node_interp(const orc_fab* c, int r, int i, int j, int k)
{
    const int f[3] = {i, j, k};
    int c0[3]; double w[3];
    for (int d = 0; d < 3; ++d) {
        c0[d] = f[d] >= 0 ? f[d] / r : -((-f[d] + r - 1) / r);
        w[d] = (double)(f[d] - c0[d] * r) / (double)r;
    }
    double v = 0.0;
    for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
        const double ww = (cx ? w[0] : 1.0 - w[0]) * (cy ? w[1] : 1.0 - w[1]) * (cz ? w[2] : 1.0 - w[2]);
        if (ww != 0.0) v += ww * A4(c, c0[0] + cx, c0[1] + cy, c0[2] + cz, 0);
    }
    return v;
}

/* Projection::set_outflow_bcs / set_outflow_bcs_at_level / computeRhoG (Projection.cpp:1721-2370), 3-D: hydrostatic pressure on the
 * nodes of an outflow face when gravity != 0.  z-hi: zero (nothing to do); z-lo: upstream aborts; x / y faces: on every node column of
 * the face, integrating down from the top,  rhog -= gravity * rhoExt * dz,  phi(node k) = rhog,  rhoExt = (3 rho1 - rho2) / 2
 * extrapolated to the face from the first two cells inside it (rho1, rho2: means of the two cell columns next to the node column;
 * at a domain edge of the face the density's BCRec decides: ext_dir the ghost column, foextrap the first column, hoextrap extrapolated).
 * Applied only where the level covers the whole two-cell strip along the face (:1776-1803).  rho: 1 filled ghost cell.
 * Upstream's y-hi branch differs from the other three as written: rho2 of the regular columns is the mean of rho(i, j-1) and
 * rho(i-1, j-2) (:2303-2304), followed here; its ext_dir low-edge column reads outside the strip (:2317-2318), refused here. */
void ns_set_outflow_bcs(const orc_ns_state* s, orc_fab* phi, const orc_fab* rho)
{
    const orc_geom* g = &s->g;
    const double grav = s->p.gravity;
    if (!(fabs(grav) > 0.0)) return;
    const int nz = g->n[2];
    const double dh = g->dx[2];
    for (int D = 0; D < 3; ++D) for (int side = 0; side < 2; ++side) {
        if (g->periodic[D] || (side == 0 ? s->p.phys_lo[D] : s->p.phys_hi[D]) != PHYS_OUTFLOW) continue;
        if (D == 2) {
            if (side == 1) continue;
            fprintf(stderr, "orc set_outflow_bcs: outflow at the bottom with gravity (Projection::computeRhoG aborts)\n"); abort();
        }
        const int T = 1 - D, nD = g->n[D], nT = g->n[T];
        /* the level must cover the whole strip (two cells deep) */
        if (s->level > 0) {
            int all = 1, any = 0;
            for (int k = 0; k < nz && (all || !any); ++k) for (int t = 0; t < nT; ++t) for (int a = 1; a <= 2; ++a) {
                int q[3]; q[D] = side == 0 ? a - 1 : nD - a; q[T] = t; q[2] = k;
                if (A4(&s->cov, q[0], q[1], q[2], 0) != 0.0) any = 1; else all = 0;
            }
            if (!all) continue;
        }
#define RS(a, t, k) (D == 0 ? A4(rho, side == 0 ? (a) - 1 : nD - (a), (t), (k), 0) : A4(rho, (t), side == 0 ? (a) - 1 : nD - (a), (k), 0))
        const int blo = s->bc_scal[0].lo[T], bhi = s->bc_scal[0].hi[T];
        const int edge_lo = !g->periodic[T] && (blo == ORC_BC_EXT_DIR || blo == ORC_BC_HOEXTRAP || blo == ORC_BC_FOEXTRAP);
        const int edge_hi = !g->periodic[T] && (bhi == ORC_BC_EXT_DIR || bhi == ORC_BC_HOEXTRAP || bhi == ORC_BC_FOEXTRAP);
        const int yhi = (D == 1 && side == 1);
        if (yhi && edge_lo && blo == ORC_BC_EXT_DIR) { fprintf(stderr, "orc set_outflow_bcs: y-hi outflow with x-lo inflow reads outside the strip upstream\n"); abort(); }
        for (int t = 0; t <= nT; ++t) {
            double rhog = 0.0;
            for (int k = nz - 1; k >= 0; --k) {
                double r1, r2;
                if (t == 0 && edge_lo) {
                    if (blo == ORC_BC_EXT_DIR) { r1 = RS(1, -1, k); r2 = RS(2, -1, k); }
                    else if (blo == ORC_BC_HOEXTRAP) { r1 = 0.5 * (3. * RS(1, 0, k) - RS(1, 1, k)); r2 = 0.5 * (3. * RS(2, 0, k) - RS(2, 1, k)); }
                    else { r1 = RS(1, 0, k); r2 = RS(2, 0, k); }
                } else if (t == nT && edge_hi) {
                    if (bhi == ORC_BC_EXT_DIR) { r1 = RS(1, nT, k); r2 = RS(2, nT, k); }
                    else if (bhi == ORC_BC_HOEXTRAP) { r1 = 0.5 * (3. * RS(1, nT - 1, k) - RS(1, nT - 2, k)); r2 = 0.5 * (3. * RS(2, nT - 1, k) - RS(2, nT - 2, k)); }
                    else { r1 = RS(1, nT - 1, k); r2 = RS(2, nT - 1, k); }
                } else {
                    r1 = 0.5 * (RS(1, t, k) + RS(1, t - 1, k));
                    r2 = yhi ? 0.5 * (RS(1, t, k) + RS(2, t - 1, k)) : 0.5 * (RS(2, t, k) + RS(2, t - 1, k));
                }
                const double rhoExt = 0.5 * (3. * r1 - r2);
                rhog -= grav * rhoExt * dh;
                int q[3]; q[D] = side == 0 ? 0 : nD; q[T] = t; q[2] = k;
                A4(phi, q[0], q[1], q[2], 0) = 0.0 + rhog;
            }
            int q[3]; q[D] = side == 0 ? 0 : nD; q[T] = t; q[2] = nz;
            A4(phi, q[0], q[1], q[2], 0) = 0.0;
        }
#undef RS
    }
}

/* Projection::level_project (Projection.cpp:166-450) */
static void level_project(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_fab* Un = S_NEW(s);
    orc_fab* Pn = P_NEW(s);
    const orc_fab* Gp = GP_OLD(s);
    if (s->level == 0) {
        /* zero P_new on the valid nodal box (level 0: nGrow 0) */
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) A4(Pn, i, j, k, 0) = 0.0;
    } else {
        /* :232-256: FillCoarsePatch(P_new, cur_pres_time) -- Press_Type is an Interval type, cur_pres_time lies in the coarse level's
         * NEW interval (the coarse level has already advanced), node_bilinear_interp -- then zero on every box shrunk by one node:
         * the nodes on the box faces keep the interpolated coarse pressure (Dirichlet data on the coarse/fine boundary, the initial
         * guess on faces shared by two boxes) */
        const orc_ns_state* c = s->crse;
        const double tp = 0.5 * (s->pt_new[0] + s->pt_new[1]);
        const double teps = 1.e-3 * fabs(c->pt_new[0] - c->pt_old[0]);
        const orc_fab* Pc;
        if (tp >= c->pt_new[0] - teps && tp <= c->pt_new[1] + teps) Pc = P_NEW(c);
        else if (tp >= c->pt_old[0] - teps && tp <= c->pt_old[1] + teps) Pc = P_OLD(c);
        else { fprintf(stderr, "orc level_project: no coarse pressure at time %g\n", tp); abort(); }
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
            A4(Pn, i, j, k, 0) = node_interp(Pc, s->ratio, i, j, k);
        for (int b = 0; b < s->nbox; ++b) {
            const int* bx = s->boxes + 6 * b;
            for (int k = bx[2] + 1; k <= bx[5]; ++k) for (int j = bx[1] + 1; j <= bx[4]; ++j) for (int i = bx[0] + 1; i <= bx[3]; ++i) A4(Pn, i, j, k, 0) = 0.0;
        }
    }
    const double dt_inv = 1. / dt;
    for (int n = 0; n < 3; ++n)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i)
        A4(Un, i, j, k, n) *= dt_inv;
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(Un, i, j, k, n) += A4(Gp, i, j, k, n) / A4(&s->rho_half, i, j, k, 0);
    ns_set_outflow_bcs(s, Pn, &s->rho_half);                     /* Projection.cpp:308-325 (LEVEL_PROJ) */
    /* scaleVar: sigma = 1/rho_half */
    orc_fab sig = orc_alloc(g->n, ORC_CELL, 1, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(&sig, i, j, k, 0) = 1.0 / A4(&s->rho_half, i, j, k, 0);
    orc_fill_periodic(&sig, g, ORC_CELL);
    orc_fab v = vel_view(Un);
    orc_fab rhcc; rhcc.p = NULL;
    if (s->have_divu) {                     /* divusource = getDivCond(1, time + dt) / dt, rhcc = -divusource (Projection.cpp:267-276, 379-389) */
        rhcc = orc_alloc(g->n, ORC_CELL, 0, 1);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&rhcc, i, j, k, 0) = -(A4(S_NEW(s), i, j, k, s->Divu) * dt_inv);
    }
    nodal_project_level(s, &v, Pn, &sig, 0, 1.0 / dt, 1, rhcc.p ? &rhcc : NULL);
    if (rhcc.p) orc_free(&rhcc);
    orc_free(&sig);
    for (int n = 0; n < 3; ++n)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i)
        A4(Un, i, j, k, n) *= dt;
}

void ns_make_rho_curr_time(orc_ns_state* s)
{
    orc_fab r = fillpatch(s, S_NEW(s), Density, 1, 1, &s->bc_scal[0]);
    orc_copy_all(&s->rho_ctime, &r);
    orc_free(&r);
}

/* NavierStokes::advance (NavierStokes.cpp:543-691) */
double ns_advance(orc_ns_state* s, double dt, int iteration, int ncycle)
{
    const orc_geom* g = &s->g;
    advance_setup(s, dt, iteration, ncycle);
    double dt_test = predict_velocity(s, dt);
    mac_project(s, dt);
    /* NavierStokes.cpp:606-623: with do_mom_diff the reference calls velocity_advection after the density update; it reads only
     * time-n data and the MAC velocities, so the result does not depend on that order */
    velocity_advection(s, dt);
    scalar_advection(s, dt);
    scalar_update_rho(s, dt);
    scalar_update_tracers(s, dt);
    scalar_diffusion_update(s, dt);
    if (s->have_divu) {                     /* NavierStokes.cpp:631-641 */
        ns_calc_divu(s, 1);
        ns_calc_dsdt(s, dt);
        if (s->initial_step)
            for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(S_OLD(s), i, j, k, s->Dsdt) = A4(S_NEW(s), i, j, k, s->Dsdt);
    }
    velocity_advection_update(s, dt);
    if (!s->initial_iter) velocity_diffusion_update(s, dt);
    else initial_velocity_diffusion_update(s, dt);
    if (!s->initial_step) {
        if (s->level > 0) {              /* incrRhoAvg((iteration==ncycle ? 0.5 : 1.0) / ncycle), :644-645 */
            const double alpha = (iteration == ncycle ? 0.5 : 1.0) / (double)ncycle;
            for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
                A4(&s->rho_avg, i, j, k, 0) += alpha * A4(S_NEW(s), i, j, k, Density);
        }
        level_project(s, dt);
        if (s->level > 0 && iteration == 1) orc_setval(&s->p_avg, 0.0);      /* :670-671 */
    }
    return dt_test;
}

static double advance(orc_ns_state* s, double dt)
{
    return ns_advance(s, dt, 1, 1);
}

/* Projection::initialSyncProject, single level */
static void initial_sync_project(orc_ns_state* s, double dt)
{
    const orc_geom* g = &s->g;
    orc_fab* phi = P_OLD(s);
    orc_setval(phi, 0.0);
    orc_fab *Un = S_NEW(s), *Uo = S_OLD(s);
    const double dt_inv = 1. / dt;
    /* ConvertUnew: u_new = (u_new - u_old)/dt */
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(Un, i, j, k, n) = (A4(Un, i, j, k, n) - A4(Uo, i, j, k, n)) * dt_inv;
    orc_fab sig = orc_alloc(g->n, ORC_CELL, 1, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(&sig, i, j, k, 0) = 1.0 / A4(&s->rho_half, i, j, k, 0);
    orc_fill_periodic(&sig, g, ORC_CELL);
    orc_fab v = vel_view(Un);
    orc_fab rhcc; rhcc.p = NULL;
    if (s->have_divu) {                     /* rhcc = -(divu(strt_time + dt) - divu(strt_time)) / dt, Projection.cpp:1008-1075, 1142-1148 */
        rhcc = orc_alloc(g->n, ORC_CELL, 0, 1);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
            A4(&rhcc, i, j, k, 0) = -((A4(S_NEW(s), i, j, k, s->Divu) - A4(S_OLD(s), i, j, k, s->Divu)) * dt_inv);
    }
    nodal_project_level(s, &v, phi, &sig, 1, 0.0, 0, rhcc.p ? &rhcc : NULL);
    if (rhcc.p) orc_free(&rhcc);
    orc_free(&sig);
    orc_fab* Pn = P_NEW(s);
    size_t N = orc_npts(Pn);
    for (size_t q = 0; q < N; ++q) Pn->p[q] += phi->p[q];
}

/* Projection::initialPressureProject (Projection.cpp:841-960; called from NavierStokesBase::post_init_state, NavierStokesBase.cpp:2416-2426,
 * whenever gravity is set): project (0,0,g) with sigma = 1/rho to establish the hydrostatic pressure; P and Gradp, old = new */
static void initial_pressure_project(orc_ns_state* s)
{
    const orc_geom* g = &s->g;
    if (!(fabs(s->p.gravity) > 0.0)) return;
    orc_fab sig = orc_alloc(g->n, ORC_CELL, 1, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(&sig, i, j, k, 0) = 1.0 / A4(S_NEW(s), i, j, k, Density);
    orc_fill_periodic(&sig, g, ORC_CELL);
    orc_fab vel = orc_alloc(g->n, ORC_CELL, 1, 3);
    { const size_t N = orc_npts(&vel); for (size_t q = 0; q < N; ++q) vel.p[q + 2 * N] = s->p.gravity; }
    {   /* Projection.cpp:855-905: sig = FillBoundary'ed + physical-BC'ed new density (1 ghost), set_outflow_bcs(INITIAL_PRESS) */
        orc_fab rho = fillpatch(s, S_NEW(s), Density, 1, 1, &s->bc_scal[0]);
        ns_set_outflow_bcs(s, P_NEW(s), &rho);
        orc_free(&rho);
    }
    nodal_project_level(s, &vel, P_NEW(s), &sig, 0, 0.0, 0, NULL);
    orc_copy_all(P_OLD(s), P_NEW(s));
    orc_copy_all(GP_OLD(s), GP_NEW(s));
    orc_free(&sig); orc_free(&vel);
}

void orc_ns_post_init(orc_ns_state* s, double stop_time)
{
    s->stop_time = stop_time;
    if (s->have_divu) {                     /* NavierStokes::initData, NavierStokes.cpp:457-479: rho at both times, divu of the initial data, dsdt = 0 */
        const orc_geom* g = &s->g;
        orc_fab r = fillpatch(s, S_NEW(s), Density, 1, 1, &s->bc_scal[0]);
        orc_copy_all(&s->rho_ctime, &r); orc_copy_all(&s->rho_ptime, &r);
        orc_free(&r);
        ns_calc_divu(s, 1);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(S_NEW(s), i, j, k, s->Dsdt) = 0.0;
    }
    /* post_init_state */
    initial_velocity_project(s);
    initial_pressure_project(s);
    s->initial_step = 1;
    /* post_init_estDT: dt = init_shrink * estTimeStep, limited by stop_time */
    double dt_init = s->p.init_shrink * ns_est_time_step(s);
    if (stop_time >= 0.0) {
        const double eps = 0.0001 * dt_init;
        if (s->time + dt_init > stop_time - eps) dt_init = stop_time - s->time;
    }
    s->dt = dt_init;
    /* post_init_press */
    if (s->p.init_iter > 0) {
        s->initial_iter = 1;
        for (int iter = 0; iter < s->p.init_iter; ++iter) {
            advance(s, dt_init);
            initial_sync_project(s, dt_init);
            /* resetState: state swapped back (new <- initial data), P/Gp: old := new; Dsdt_Type is not reset ("we want to improve dsdt with
             * press iters", NavierStokesBase.cpp:2669-2676): the computed dsdt stays the new data */
            s->inew = 1 - s->inew;
            if (s->have_divu) {
                const orc_geom* g = &s->g;
                for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(S_NEW(s), i, j, k, s->Dsdt) = A4(S_OLD(s), i, j, k, s->Dsdt);
            }
            orc_copy_all(P_OLD(s), P_NEW(s));
            orc_copy_all(GP_OLD(s), GP_NEW(s));
            s->initial_iter = 0;
        }
    }
    s->initial_step = 0;
    s->dt_min_adv = 1.e200;
}

double orc_ns_step(orc_ns_state* s)
{
    double dt = s->dt;
    if (s->nstep > 0) {
        /* computeNewDt */
        double dt_min = fmin(s->dt_min_adv, ns_est_time_step(s));
        if (s->p.fixed_dt <= 0.0) dt_min = fmin(dt_min, s->p.change_max * s->dt);
        dt = dt_min;
        if (s->stop_time >= 0.0) {                               /* computeNewDt, NavierStokesBase.cpp:1008-1015 */
            const double eps = 0.0001 * dt;
            if (s->time + dt > s->stop_time - eps) dt = s->stop_time - s->time;
        }
    }
    s->dt = dt;
    s->dt_min_adv = advance(s, dt);
    s->time += dt;
    s->nstep += 1;
    return dt;
}
