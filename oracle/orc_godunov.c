/* oracle/orc_godunov.c -- Godunov (PLM, corner-transport-upwind) extrapolation and advective
 * update, restated on the CPU as a multi-pass, array-at-a-time algorithm (test infrastructure
 * only; PARITY UNPINNED, see orc.h).
 *
 * Follows (upstream AMReX-Hydro, not in /root/reference): Godunov::ExtrapVelToFaces,
 * ComputeAdvectiveVel, ExtrapVelToFacesOnBox, Godunov::ComputeEdgeState, PLM::PredictVelOnXFace /
 * PredictStateOnXFace, amrex_calc_xslope_extdir (order 4), Godunov_corner_couple_*,
 * SetTransTermXBCs / SetXEdgeBCs, HydroUtils::ComputeFluxes / ComputeDivergence /
 * ComputeConvectiveTerm.  Reference call sites: Source/NavierStokesBase.cpp:4487-4491
 * (ExtrapVelToFaces) and :4701-4842 (ComputeAofs: area-weighted fluxes, mult=-1,
 * convective correction, aofs = -update).
 *
 * Index conventions: cell c in direction d has low face c and high face c+1.
 */
#include "orc_int.h"

#define SMALL_VEL 1.e-8

static inline double Q(const orc_fab* f, const int c[3], int n) { return A4(f, c[0], c[1], c[2], n); }
static inline double* QP(orc_fab* f, const int c[3], int n) { return &A4(f, c[0], c[1], c[2], n); }
static inline void shift(int o[3], const int c[3], int d, int s) { o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[d] += s; }

/* limited 2nd-order slope ingredient (one cell) */
static inline double lim2(double dlft, double drgt)
{
    double dcen = 0.5 * (dlft + drgt);
    double dsgn = copysign(1.0, dcen);
    double dlim = (dlft * drgt >= 0.0) ? 2.0 * fmin(fabs(dlft), fabs(drgt)) : 0.0;
    return dsgn * fmin(dlim, fabs(dcen));
}

/* amrex_calc_xslope_extdir, order 4, direction dir.  edlo/edhi: ext_dir (or hoextrap) at the
 * low/high domain face; domlo/domhi: first/last interior cell index in direction dir. */
static double slope4_extdir(const orc_fab* q, const int c[3], int n, int dir, int edlo, int edhi, int domlo, int domhi)
{
    int m1[3], m2[3], p1[3], p2[3];
    shift(m1, c, dir, -1); shift(m2, c, dir, -2); shift(p1, c, dir, 1); shift(p2, c, dir, 2);
    const double qi = Q(q, c, n), qm = Q(q, m1, n), qp = Q(q, p1, n), qmm = Q(q, m2, n), qpp = Q(q, p2, n);
    double dfm = lim2(qm - qmm, qi - qm);
    double dfp = lim2(qp - qi, qpp - qp);
    double dlft = qi - qm, drgt = qp - qi;
    double dcen = 0.5 * (dlft + drgt);
    double dsgn = copysign(1.0, dcen);
    double dlim = (dlft * drgt >= 0.0) ? 2.0 * fmin(fabs(dlft), fabs(drgt)) : 0.0;
    double dtemp = 4.0 / 3.0 * dcen - 1.0 / 6.0 * (dfp + dfm);
    const int i = c[dir];
    if (edlo && i == domlo) {
        dtemp = -16. / 15. * qm + .5 * qi + 2. / 3. * qp - 0.1 * qpp;
        dlft = 2. * (qi - qm); drgt = 2. * (qp - qi);
        dlim = (dlft * drgt >= 0.0) ? fmin(fabs(dlft), fabs(drgt)) : 0.0;
        dsgn = copysign(1.0, dtemp);
    } else if (edlo && i == domlo + 1) {
        /* slope of cell domlo recomputed with the one-sided formula */
        dfm = -16. / 15. * qmm + .5 * qm + 2. / 3. * qi - 0.1 * qp;
        double l = 2. * (qm - qmm), r = 2. * (qi - qm);
        double dlimsh = (l * r >= 0.0) ? fmin(fabs(l), fabs(r)) : 0.0;
        double dsgnsh = copysign(1.0, dfm);
        dfm = dsgnsh * fmin(dlimsh, fabs(dfm));
        dtemp = 4.0 / 3.0 * dcen - 1.0 / 6.0 * (dfp + dfm);
    }
    if (edhi && i == domhi) {
        dtemp = 16. / 15. * qp - .5 * qi - 2. / 3. * qm + 0.1 * qmm;
        dlft = 2. * (qi - qm); drgt = 2. * (qp - qi);
        dlim = (dlft * drgt >= 0.0) ? fmin(fabs(dlft), fabs(drgt)) : 0.0;
        dsgn = copysign(1.0, dtemp);
    } else if (edhi && i == domhi - 1) {
        dfp = 16. / 15. * qpp - .5 * qp - 2. / 3. * qi + 0.1 * qm;
        double l = 2. * (qp - qi), r = 2. * (qpp - qp);
        double dlimsh = (l * r >= 0.0) ? fmin(fabs(l), fabs(r)) : 0.0;
        double dsgnsh = copysign(1.0, dfp);
        dfp = dsgnsh * fmin(dlimsh, fabs(dfp));
        dtemp = 4.0 / 3.0 * dcen - 1.0 / 6.0 * (dfp + dfm);
    }
    return dsgn * fmin(dlim, fabs(dtemp));
}

double orc_slope4(const orc_fab* q, int i, int j, int k, int n, int dir)
{
    int c[3] = {i, j, k};
    return slope4_extdir(q, c, n, dir, 0, 0, 0, 0);
}

static inline int bc_is_ed_or_ho(int b) { return b == ORC_BC_EXT_DIR || b == ORC_BC_HOEXTRAP; }

/* ---- PPM (ns.advection_scheme = Godunov_PPM, reference Source/NavierStokesBase.cpp:548-553, 4654-4656: AMReX-Hydro "Godunov" with
 * use_ppm).  The piecewise-parabolic reconstruction of Colella & Woodward (JCP 54, 1984) as AMReX-Hydro's hydro_godunov_ppm states it
 * (upstream, not in the reference tree; restated from the published algorithm):
 *   van Leer slopes  dvl(c) = sign(dc) min(|dc|, 2|s_c - s_{c-1}|, 2|s_{c+1} - s_c|) (0 at an extremum), dc = (s_{c+1} - s_{c-1})/2
 *   edge values      s_{c+1/2} = (s_c + s_{c+1})/2 - (dvl(c+1) - dvl(c))/6, clipped to [min, max] of the two cells
 *   ext_dir / hoextrap face at the low end: the first cell takes the boundary value on its low edge and the one-sided 4-point value
 *     -1/5 s_b + 3/4 s_0 + 1/2 s_1 - 1/20 s_2 (clipped to cells 0, 1) on its high edge, which is also the low edge of the second cell
 *   monotonisation   (sp - s)(s - sm) <= 0 -> sp = sm = s;  |sp - s| >= 2|sm - s| -> sp = 3s - 2sm;  |sm - s| >= 2|sp - s| -> sm = 3s - 2sp
 *   tracing          s6 = 6s - 3(sm + sp), sigma = |u| dt/dx;
 *                    high face: u > small ? sp - sigma/2 ((sp - sm) - (1 - 2/3 sigma) s6) : s
 *                    low  face: u < -small ? sm + sigma/2 ((sp - sm) + (1 - 2/3 sigma) s6) : s */
/* the value is ns.advection_scheme: 0 Godunov_PLM, 1 Godunov_PPM, 2 BDS (edge states of ComputeAofs by orc_bds.c; the velocity
 * prediction stays Godunov_PLM: NavierStokesBase.cpp:4487 passes godunov_use_ppm = (advection_scheme == "Godunov_PPM")) */
static int g_use_ppm = 0, g_bds = 0;
void orc_godunov_set_ppm(int scheme) { g_use_ppm = scheme == 1; g_bds = scheme == 2; }
int orc_godunov_get_ppm(void) { return g_bds ? 2 : g_use_ppm; }
void orc_bds_edge_state(const orc_geom* g, const orc_fab* q, int ncomp, const orc_fab* fq, orc_fab* const mac[3], const int* iconserv,
                        double dt, const orc_bcrec* bc, int is_velocity, orc_fab* edge);

static inline double vanleer(double s0, double sm1, double sp1)
{
    const double dsc = 0.5 * (sp1 - sm1), dsl = 2.0 * (s0 - sm1), dsr = 2.0 * (sp1 - s0);
    return (dsl * dsr > 0.0) ? copysign(1.0, dsc) * fmin(fabs(dsc), fmin(fabs(dsl), fabs(dsr))) : 0.0;
}
static inline double clip2(double v, double a, double b) { return fmax(fmin(v, fmax(a, b)), fmin(a, b)); }
static inline double qd(const orc_fab* q, const int c[3], int n, int d, int o) { int e[3] = {c[0], c[1], c[2]}; e[d] += o; return A4(q, e[0], e[1], e[2], n); }

/* monotonised edge values of cell c in direction d */
static void ppm_edges(const orc_fab* q, const int c[3], int n, int d, int edlo, int edhi, int domlo, int domhi, double* smo, double* spo)
{
    const double s0 = qd(q, c, n, d, 0), sm1 = qd(q, c, n, d, -1), sm2 = qd(q, c, n, d, -2), sp1 = qd(q, c, n, d, 1), sp2 = qd(q, c, n, d, 2);
    const double dm = vanleer(sm1, sm2, s0), d0 = vanleer(s0, sm1, sp1), dp = vanleer(sp1, s0, sp2);
    double sm = clip2(0.5 * (s0 + sm1) - (1.0 / 6.0) * (d0 - dm), s0, sm1);
    double sp = clip2(0.5 * (sp1 + s0) - (1.0 / 6.0) * (dp - d0), sp1, s0);
    const int i = c[d];
    if (edlo && (i == domlo || i == domlo + 1)) {
        const int o = domlo - i;        /* offset of cell domlo from c */
        const double sb = qd(q, c, n, d, o - 1), a0 = qd(q, c, n, d, o), a1 = qd(q, c, n, d, o + 1), a2 = qd(q, c, n, d, o + 2);
        const double se = clip2(-0.2 * sb + 0.75 * a0 + 0.5 * a1 - 0.05 * a2, a1, a0);
        if (i == domlo) { sm = sb; sp = se; } else sm = se;
    }
    if (edhi && (i == domhi || i == domhi - 1)) {
        const int o = domhi - i;
        const double sb = qd(q, c, n, d, o + 1), a0 = qd(q, c, n, d, o), a1 = qd(q, c, n, d, o - 1), a2 = qd(q, c, n, d, o - 2);
        const double se = clip2(-0.2 * sb + 0.75 * a0 + 0.5 * a1 - 0.05 * a2, a1, a0);
        if (i == domhi) { sp = sb; sm = se; } else sp = se;
    }
    if ((sp - s0) * (s0 - sm) <= 0.0) { sp = s0; sm = s0; }
    else if (fabs(sp - s0) >= 2.0 * fabs(sm - s0)) sp = 3.0 * s0 - 2.0 * sm;
    else if (fabs(sm - s0) >= 2.0 * fabs(sp - s0)) sm = 3.0 * s0 - 2.0 * sp;
    *smo = sm; *spo = sp;
}
#define PPM_SMALL_VEL 1.e-8
/* state on the high (side = 1) / low (side = 0) face of cell c traced with velocity u */
static double ppm_trace(const orc_fab* q, const int c[3], int n, int d, int edlo, int edhi, int domlo, int domhi, double u, double dtdx, int side)
{
    double sm, sp;
    ppm_edges(q, c, n, d, edlo, edhi, domlo, domhi, &sm, &sp);
    const double s0 = Q(q, c, n), s6 = 6.0 * s0 - 3.0 * (sm + sp), sigma = fabs(u) * dtdx;
    if (side) return (u > PPM_SMALL_VEL) ? sp - (0.5 * sigma) * ((sp - sm) - (1.0 - (2.0 / 3.0) * sigma) * s6) : s0;
    return (u < -PPM_SMALL_VEL) ? sm + (0.5 * sigma) * ((sp - sm) + (1.0 - (2.0 / 3.0) * sigma) * s6) : s0;
}

/* SetTransTerm{X,Y,Z}BCs: face f (index in direction d), states lo (from cell f-1) and hi (cell f) */
static void trans_bc(const orc_fab* q, const int fidx[3], int n, int d, double* lo, double* hi,
                     int bclo, int bchi, int domlo, int domhi, int is_velocity)
{
    const int f = fidx[d];
    if (f <= domlo) {
        if (bclo == ORC_BC_EXT_DIR) {
            int c[3] = {fidx[0], fidx[1], fidx[2]}; c[d] = domlo - 1;
            *lo = Q(q, c, n);
            if (n == d && is_velocity) *hi = *lo;
        } else if (bclo == ORC_BC_FOEXTRAP || bclo == ORC_BC_HOEXTRAP || bclo == ORC_BC_REFLECT_EVEN) {
            *lo = *hi;
        } else if (bclo == ORC_BC_REFLECT_ODD) {
            *hi = 0.; *lo = 0.;
        }
    } else if (f > domhi) {
        if (bchi == ORC_BC_EXT_DIR) {
            int c[3] = {fidx[0], fidx[1], fidx[2]}; c[d] = domhi + 1;
            *hi = Q(q, c, n);
            if (n == d && is_velocity) *lo = *hi;
        } else if (bchi == ORC_BC_FOEXTRAP || bchi == ORC_BC_HOEXTRAP || bchi == ORC_BC_REFLECT_EVEN) {
            *hi = *lo;
        } else if (bchi == ORC_BC_REFLECT_ODD) {
            *lo = 0.; *hi = 0.;
        }
    }
}

/* Set{X,Y,Z}EdgeBCs: final edge states on the domain faces */
static void edge_bc(const orc_fab* q, const int fidx[3], int n, int d, double* lo, double* hi,
                    int bclo, int bchi, int domlo, int domhi, int is_velocity)
{
    const int f = fidx[d];
    if (f <= domlo) {
        if (bclo == ORC_BC_EXT_DIR) {
            int c[3] = {fidx[0], fidx[1], fidx[2]}; c[d] = domlo - 1;
            *lo = Q(q, c, n);
            if (n == d && is_velocity) *hi = *lo;
        } else if (bclo == ORC_BC_FOEXTRAP || bclo == ORC_BC_HOEXTRAP || bclo == ORC_BC_REFLECT_EVEN) {
            if (n == d && is_velocity && bclo != ORC_BC_REFLECT_EVEN) *hi = fmin(*hi, 0.);
            *lo = *hi;
        } else if (bclo == ORC_BC_REFLECT_ODD) {
            *hi = 0.; *lo = 0.;
        }
    } else if (f > domhi) {
        if (bchi == ORC_BC_EXT_DIR) {
            int c[3] = {fidx[0], fidx[1], fidx[2]}; c[d] = domhi + 1;
            *hi = Q(q, c, n);
            if (n == d && is_velocity) *lo = *hi;
        } else if (bchi == ORC_BC_FOEXTRAP || bchi == ORC_BC_HOEXTRAP || bchi == ORC_BC_REFLECT_EVEN) {
            if (n == d && is_velocity && bchi != ORC_BC_REFLECT_EVEN) *lo = fmax(*lo, 0.);
            *hi = *lo;
        } else if (bchi == ORC_BC_REFLECT_ODD) {
            *lo = 0.; *hi = 0.;
        }
    }
}

/* face-array allocation: faces of direction d over cells grown by gt in the transverse directions */
static orc_fab alloc_faces(const orc_geom* g, int d, int gt, int nc)
{
    orc_fab f;
    for (int e = 0; e < 3; ++e) { f.lo[e] = (e == d) ? 0 : -gt; f.hi[e] = (e == d) ? g->n[e] : g->n[e] - 1 + gt; }
    f.nc = nc;
    f.p = (double*)calloc(orc_npts(&f) * (size_t)nc, sizeof(double));
    return f;
}

/* every iteration declares its own index triple (the enclosing declaration of the same name is shadowed), so the planes can be
 * shared out to OpenMP threads (orc_threads, default 1; used by bench.py's cpu_baseline) without changing any result */
#define LOOP3(f, c) _Pragma("omp parallel for schedule(static) num_threads(orc_threads)") \
    for (int c##_k = (f)->lo[2]; c##_k <= (f)->hi[2]; ++c##_k) for (int c##_j = (f)->lo[1]; c##_j <= (f)->hi[1]; ++c##_j) \
    for (int c##_i = (f)->lo[0]; c##_i <= (f)->hi[0]; ++c##_i) for (int c[3] = {c##_i, c##_j, c##_k}, c##_once = 1; c##_once; c##_once = 0)

/* PLM trace: Im[d](c) = state at the low face of cell c, Ip[d](c) = state at the high face of cell c.
 * trace velocity: cell-centred vcc(c, d) (predict) or face-centred umac (advect, both sides use the
 * face's own umac -- handled by the caller passing a per-face functor is avoided by splitting). */
static void plm_predict_vel(const orc_geom* g, const orc_fab* q, int ncomp, const orc_fab* vcc,
                            orc_fab Im[3], orc_fab Ip[3], double dt, const orc_bcrec* bc)
{
    for (int d = 0; d < 3; ++d) {
        const double dtdx = dt / g->dx[d];
        for (int n = 0; n < ncomp; ++n) {
            const int edlo = !g->periodic[d] && bc_is_ed_or_ho(bc[n].lo[d]);
            const int edhi = !g->periodic[d] && bc_is_ed_or_ho(bc[n].hi[d]);
            LOOP3(&Im[d], c) {
                double u = Q(vcc, c, d);
                if (g_use_ppm) {
                    *QP(&Im[d], c, n) = ppm_trace(q, c, n, d, edlo, edhi, 0, g->n[d] - 1, u, dtdx, 0);
                    *QP(&Ip[d], c, n) = ppm_trace(q, c, n, d, edlo, edhi, 0, g->n[d] - 1, u, dtdx, 1);
                    continue;
                }
                double sl = slope4_extdir(q, c, n, d, edlo, edhi, 0, g->n[d] - 1);
                *QP(&Im[d], c, n) = Q(q, c, n) + 0.5 * (-1.0 - u * dtdx) * sl;
                *QP(&Ip[d], c, n) = Q(q, c, n) + 0.5 * (1.0 - u * dtdx) * sl;
            }
        }
    }
}

void orc_extrap_vel_to_faces(const orc_geom* g, const orc_fab* vel, const orc_fab* force, orc_fab* umac[3],
                             double dt, const orc_bcrec* bc, int use_forces_in_trans)
{
    const int ncomp = 3;
    orc_fab Im[3], Ip[3], ad[3], lo_[3], hi_[3], edge[3];
    for (int d = 0; d < 3; ++d) {
        Im[d] = orc_alloc(g->n, ORC_CELL, 1, ncomp);
        Ip[d] = orc_alloc(g->n, ORC_CELL, 1, ncomp);
        ad[d] = alloc_faces(g, d, 1, 1);
        lo_[d] = alloc_faces(g, d, 1, ncomp);
        hi_[d] = alloc_faces(g, d, 1, ncomp);
        edge[d] = alloc_faces(g, d, 1, ncomp);
    }
    plm_predict_vel(g, vel, ncomp, vel, Im, Ip, dt, bc);

    /* ComputeAdvectiveVel: normal component only */
    for (int d = 0; d < 3; ++d) {
        LOOP3(&ad[d], f) {
            int cm[3]; shift(cm, f, d, -1);
            double lo = Q(&Ip[d], cm, d), hi = Q(&Im[d], f, d);
            if (use_forces_in_trans && force) { lo += 0.5 * dt * Q(force, cm, d); hi += 0.5 * dt * Q(force, f, d); }
            if (!g->periodic[d]) trans_bc(vel, f, d, d, &lo, &hi, bc[d].lo[d], bc[d].hi[d], 0, g->n[d] - 1, 1);
            double st = ((lo + hi) >= 0.) ? lo : hi;
            int ltm = ((lo <= 0. && hi >= 0.) || (fabs(lo + hi) < SMALL_VEL));
            *QP(&ad[d], f, 0) = ltm ? 0. : st;
        }
    }
    /* upwind every component with the advective velocity */
    for (int d = 0; d < 3; ++d) {
        for (int n = 0; n < ncomp; ++n)
        LOOP3(&edge[d], f) {
            int cm[3]; shift(cm, f, d, -1);
            double lo = Q(&Ip[d], cm, n), hi = Q(&Im[d], f, n);
            if (use_forces_in_trans && force) { lo += 0.5 * dt * Q(force, cm, n); hi += 0.5 * dt * Q(force, f, n); }
            double uad = Q(&ad[d], f, 0);
            if (!g->periodic[d]) trans_bc(vel, f, n, d, &lo, &hi, bc[n].lo[d], bc[n].hi[d], 0, g->n[d] - 1, 1);
            *QP(&lo_[d], f, n) = lo; *QP(&hi_[d], f, n) = hi;
            double st = (uad >= 0.) ? lo : hi;
            double fu = (fabs(uad) < SMALL_VEL) ? 0.0 : 1.0;
            *QP(&edge[d], f, n) = fu * st + (1.0 - fu) * 0.5 * (hi + lo);
        }
    }
    /* final states: for normal direction d, component n = d */
    for (int d = 0; d < 3; ++d) {
        const int n = d;
        /* T[t]: state on t-faces corrected with the derivative in the other transverse direction o */
        orc_fab T[3];
        for (int t = 0; t < 3; ++t) {
            if (t == d) { T[t].p = NULL; continue; }
            const int o = 3 - d - t;
            /* t-faces over cells: grown 1 in d, 0 in o */
            orc_fab* Tt = &T[t];
            for (int e = 0; e < 3; ++e) {
                if (e == t) { Tt->lo[e] = 0; Tt->hi[e] = g->n[e]; }
                else if (e == d) { Tt->lo[e] = -1; Tt->hi[e] = g->n[e]; }
                else { Tt->lo[e] = 0; Tt->hi[e] = g->n[e] - 1; }
            }
            Tt->nc = 1;
            Tt->p = (double*)calloc(orc_npts(Tt), sizeof(double));
            LOOP3(Tt, f) {
                int cm[3]; shift(cm, f, t, -1);       /* cell on the low side of the t-face */
                int cmo[3], fo[3];
                shift(cmo, cm, o, 1); shift(fo, f, o, 1);
                /* Godunov_corner_couple (non-conservative form) */
                double l = Q(&lo_[t], f, n) - dt / (6.0 * g->dx[o]) * (Q(&ad[o], cmo, 0) + Q(&ad[o], cm, 0)) * (Q(&edge[o], cmo, n) - Q(&edge[o], cm, n));
                double h = Q(&hi_[t], f, n) - dt / (6.0 * g->dx[o]) * (Q(&ad[o], fo, 0) + Q(&ad[o], f, 0)) * (Q(&edge[o], fo, n) - Q(&edge[o], f, n));
                double tad = Q(&ad[t], f, 0);
                if (!g->periodic[t]) trans_bc(vel, f, n, t, &l, &h, bc[n].lo[t], bc[n].hi[t], 0, g->n[t] - 1, 1);
                double st = (tad >= 0.) ? l : h;
                double fu = (fabs(tad) < SMALL_VEL) ? 0.0 : 1.0;
                *QP(Tt, f, 0) = fu * st + (1.0 - fu) * 0.5 * (h + l);
            }
        }
        int f[3];
        for (f[2] = 0; f[2] <= g->n[2] - 1 + (d == 2); ++f[2])
        for (f[1] = 0; f[1] <= g->n[1] - 1 + (d == 1); ++f[1])
        for (f[0] = 0; f[0] <= g->n[0] - 1 + (d == 0); ++f[0]) {
            int cm[3]; shift(cm, f, d, -1);
            double stl = Q(&lo_[d], f, n), sth = Q(&hi_[d], f, n);
            for (int t = 0; t < 3; ++t) {
                if (t == d) continue;
                int cmt[3], ft[3];
                shift(cmt, cm, t, 1); shift(ft, f, t, 1);
                stl -= (0.25 * dt / g->dx[t]) * (Q(&ad[t], cmt, 0) + Q(&ad[t], cm, 0)) * (Q(&T[t], cmt, 0) - Q(&T[t], cm, 0));
                sth -= (0.25 * dt / g->dx[t]) * (Q(&ad[t], ft, 0) + Q(&ad[t], f, 0)) * (Q(&T[t], ft, 0) - Q(&T[t], f, 0));
            }
            if (!use_forces_in_trans && force) { stl += 0.5 * dt * Q(force, cm, n); sth += 0.5 * dt * Q(force, f, n); }
            if (!g->periodic[d]) edge_bc(vel, f, n, d, &stl, &sth, bc[n].lo[d], bc[n].hi[d], 0, g->n[d] - 1, 1);
            double st = ((stl + sth) >= 0.) ? stl : sth;
            int ltm = ((stl <= 0. && sth >= 0.) || (fabs(stl + sth) < SMALL_VEL));
            *QP(umac[d], f, 0) = ltm ? 0. : st;
        }
        for (int t = 0; t < 3; ++t) if (T[t].p) orc_free(&T[t]);
    }
    for (int d = 0; d < 3; ++d) { orc_free(&Im[d]); orc_free(&Ip[d]); orc_free(&ad[d]); orc_free(&lo_[d]); orc_free(&hi_[d]); orc_free(&edge[d]); }
}

/* Godunov::ComputeEdgeState (PLM) */
static void compute_edge_state(const orc_geom* g, const orc_fab* q, int ncomp, const orc_fab* fq, const orc_fab* divu,
                               orc_fab* const umac[3], const int* iconserv, double dt, const orc_bcrec* bc,
                               int is_velocity, int use_forces_in_trans, orc_fab edge_out[3])
{
    if (g_bds) { orc_bds_edge_state(g, q, ncomp, fq, umac, iconserv, dt, bc, is_velocity, edge_out); return; }
    orc_fab Im[3], Ip[3], lo_[3], hi_[3], edge[3];
    for (int d = 0; d < 3; ++d) {
        Im[d] = orc_alloc(g->n, ORC_CELL, 1, ncomp);
        Ip[d] = orc_alloc(g->n, ORC_CELL, 1, ncomp);
        lo_[d] = alloc_faces(g, d, 1, ncomp);
        hi_[d] = alloc_faces(g, d, 1, ncomp);
        edge[d] = alloc_faces(g, d, 1, ncomp);
    }
    /* PLM::PredictStateOnXFace: both sides of a face are traced with that face's umac */
    for (int d = 0; d < 3; ++d) {
        const double dtdx = dt / g->dx[d];
        for (int n = 0; n < ncomp; ++n) {
            const int edlo = !g->periodic[d] && bc_is_ed_or_ho(bc[n].lo[d]);
            const int edhi = !g->periodic[d] && bc_is_ed_or_ho(bc[n].hi[d]);
            LOOP3(&edge[d], f) {
                int cm[3]; shift(cm, f, d, -1);
                double um = Q(umac[d], f, 0);
                double upls, umns;
                if (g_use_ppm) {       /* PPM::PredictStateOnXFace: both sides of the face traced with the face's umac */
                    upls = ppm_trace(q, f, n, d, edlo, edhi, 0, g->n[d] - 1, um, dtdx, 0);
                    umns = ppm_trace(q, cm, n, d, edlo, edhi, 0, g->n[d] - 1, um, dtdx, 1);
                } else {
                upls = Q(q, f, n) + 0.5 * (-1.0 - um * dtdx) * slope4_extdir(q, f, n, d, edlo, edhi, 0, g->n[d] - 1);
                umns = Q(q, cm, n) + 0.5 * (1.0 - um * dtdx) * slope4_extdir(q, cm, n, d, edlo, edhi, 0, g->n[d] - 1);
                }
                *QP(&Ip[d], cm, n) = umns;
                *QP(&Im[d], f, n) = upls;
            }
        }
    }
    for (int d = 0; d < 3; ++d) {
        for (int n = 0; n < ncomp; ++n)
        LOOP3(&edge[d], f) {
            int cm[3]; shift(cm, f, d, -1);
            double uad = Q(umac[d], f, 0);
            double fux = (fabs(uad) < SMALL_VEL) ? 0. : 1.;
            int uval = uad >= 0.;
            double lo = Q(&Ip[d], cm, n), hi = Q(&Im[d], f, n);
            if (use_forces_in_trans && fq) { lo += 0.5 * dt * Q(fq, cm, n); hi += 0.5 * dt * Q(fq, f, n); }
            if (!g->periodic[d]) trans_bc(q, f, n, d, &lo, &hi, bc[n].lo[d], bc[n].hi[d], 0, g->n[d] - 1, is_velocity);
            *QP(&lo_[d], f, n) = lo; *QP(&hi_[d], f, n) = hi;
            double st = uval ? lo : hi;
            *QP(&edge[d], f, n) = fux * st + (1. - fux) * 0.5 * (hi + lo);
        }
    }
    for (int d = 0; d < 3; ++d) {
        orc_fab T[3];
        for (int t = 0; t < 3; ++t) {
            if (t == d) { T[t].p = NULL; continue; }
            const int o = 3 - d - t;
            orc_fab* Tt = &T[t];
            for (int e = 0; e < 3; ++e) {
                if (e == t) { Tt->lo[e] = 0; Tt->hi[e] = g->n[e]; }
                else if (e == d) { Tt->lo[e] = -1; Tt->hi[e] = g->n[e]; }
                else { Tt->lo[e] = 0; Tt->hi[e] = g->n[e] - 1; }
            }
            Tt->nc = ncomp;
            Tt->p = (double*)calloc(orc_npts(Tt) * (size_t)ncomp, sizeof(double));
            for (int n = 0; n < ncomp; ++n)
            LOOP3(Tt, f) {
                int cm[3]; shift(cm, f, t, -1);
                int cmo[3], fo[3];
                shift(cmo, cm, o, 1); shift(fo, f, o, 1);
                double l, h;
                if (iconserv[n]) {
                    double dvl = divu ? Q(divu, cm, 0) : 0.0, dvh = divu ? Q(divu, f, 0) : 0.0;
                    l = Q(&lo_[t], f, n) - dt / (3.0 * g->dx[o]) * (Q(&edge[o], cmo, n) * Q(umac[o], cmo, 0) - Q(&edge[o], cm, n) * Q(umac[o], cm, 0))
                        + dt / 3.0 * Q(q, cm, n) * ((Q(umac[o], cmo, 0) - Q(umac[o], cm, 0)) / g->dx[o] - 0.5 * dvl);
                    h = Q(&hi_[t], f, n) - dt / (3.0 * g->dx[o]) * (Q(&edge[o], fo, n) * Q(umac[o], fo, 0) - Q(&edge[o], f, n) * Q(umac[o], f, 0))
                        + dt / 3.0 * Q(q, f, n) * ((Q(umac[o], fo, 0) - Q(umac[o], f, 0)) / g->dx[o] - 0.5 * dvh);
                } else {
                    l = Q(&lo_[t], f, n) - dt / (6.0 * g->dx[o]) * (Q(umac[o], cmo, 0) + Q(umac[o], cm, 0)) * (Q(&edge[o], cmo, n) - Q(&edge[o], cm, n));
                    h = Q(&hi_[t], f, n) - dt / (6.0 * g->dx[o]) * (Q(umac[o], fo, 0) + Q(umac[o], f, 0)) * (Q(&edge[o], fo, n) - Q(&edge[o], f, n));
                }
                double tad = Q(umac[t], f, 0);
                if (!g->periodic[t]) trans_bc(q, f, n, t, &l, &h, bc[n].lo[t], bc[n].hi[t], 0, g->n[t] - 1, is_velocity);
                double st = (tad >= 0.) ? l : h;
                double fu = (fabs(tad) < SMALL_VEL) ? 0.0 : 1.0;
                *QP(Tt, f, n) = fu * st + (1.0 - fu) * 0.5 * (h + l);
            }
        }
        int f[3];
        for (int n = 0; n < ncomp; ++n)
        for (f[2] = 0; f[2] <= g->n[2] - 1 + (d == 2); ++f[2])
        for (f[1] = 0; f[1] <= g->n[1] - 1 + (d == 1); ++f[1])
        for (f[0] = 0; f[0] <= g->n[0] - 1 + (d == 0); ++f[0]) {
            int cm[3]; shift(cm, f, d, -1);
            double stl = Q(&lo_[d], f, n), sth = Q(&hi_[d], f, n);
            if (iconserv[n]) {
                for (int t = 0; t < 3; ++t) {
                    if (t == d) continue;
                    int cmt[3], ft[3];
                    shift(cmt, cm, t, 1); shift(ft, f, t, 1);
                    stl += -(0.5 * dt / g->dx[t]) * (Q(&T[t], cmt, n) * Q(umac[t], cmt, 0) - Q(&T[t], cm, n) * Q(umac[t], cm, 0));
                    sth += -(0.5 * dt / g->dx[t]) * (Q(&T[t], ft, n) * Q(umac[t], ft, 0) - Q(&T[t], f, n) * Q(umac[t], f, 0));
                }
                for (int t = 0; t < 3; ++t) {
                    if (t == d) continue;
                    int cmt[3], ft[3];
                    shift(cmt, cm, t, 1); shift(ft, f, t, 1);
                    stl += (0.5 * dt / g->dx[t]) * Q(q, cm, n) * (Q(umac[t], cmt, 0) - Q(umac[t], cm, 0));
                    sth += (0.5 * dt / g->dx[t]) * Q(q, f, n) * (Q(umac[t], ft, 0) - Q(umac[t], f, 0));
                }
                if (divu) { stl -= 0.5 * dt * Q(q, cm, n) * Q(divu, cm, 0); sth -= 0.5 * dt * Q(q, f, n) * Q(divu, f, 0); }
            } else {
                for (int t = 0; t < 3; ++t) {
                    if (t == d) continue;
                    int cmt[3], ft[3];
                    shift(cmt, cm, t, 1); shift(ft, f, t, 1);
                    stl -= (0.25 * dt / g->dx[t]) * (Q(umac[t], cmt, 0) + Q(umac[t], cm, 0)) * (Q(&T[t], cmt, n) - Q(&T[t], cm, n));
                    sth -= (0.25 * dt / g->dx[t]) * (Q(umac[t], ft, 0) + Q(umac[t], f, 0)) * (Q(&T[t], ft, n) - Q(&T[t], f, n));
                }
            }
            if (!use_forces_in_trans && fq) { stl += 0.5 * dt * Q(fq, cm, n); sth += 0.5 * dt * Q(fq, f, n); }
            if (!g->periodic[d]) edge_bc(q, f, n, d, &stl, &sth, bc[n].lo[d], bc[n].hi[d], 0, g->n[d] - 1, is_velocity);
            double um = Q(umac[d], f, 0);
            double temp = (um >= 0.) ? stl : sth;
            temp = (fabs(um) < SMALL_VEL) ? 0.5 * (stl + sth) : temp;
            *QP(&edge_out[d], f, n) = temp;
        }
        for (int t = 0; t < 3; ++t) if (T[t].p) orc_free(&T[t]);
    }
    for (int d = 0; d < 3; ++d) { orc_free(&Im[d]); orc_free(&Ip[d]); orc_free(&lo_[d]); orc_free(&hi_[d]); orc_free(&edge[d]); }
}

void orc_compute_aofs(const orc_geom* g, orc_fab* aofs, int acomp, const orc_fab* S, int ncomp,
                      const orc_fab* force, const orc_fab* divu, orc_fab* const umac[3], const int* iconserv,
                      double dt, const orc_bcrec* bc, int is_velocity, int use_forces_in_trans,
                      orc_fab* edge_out[3], orc_fab* flux_out[3])
{
    orc_fab edge[3], flux[3];
    for (int d = 0; d < 3; ++d) { edge[d] = alloc_faces(g, d, 0, ncomp); flux[d] = alloc_faces(g, d, 0, ncomp); }
    compute_edge_state(g, S, ncomp, force, divu, umac, iconserv, dt, bc, is_velocity, use_forces_in_trans, edge);
    /* ComputeFluxes, area-weighted (NavierStokesBase.cpp:4651) */
    for (int d = 0; d < 3; ++d) {
        const double area = g->dx[(d + 1) % 3] * g->dx[(d + 2) % 3];
        for (int n = 0; n < ncomp; ++n)
        LOOP3(&flux[d], f) *QP(&flux[d], f, n) = Q(&edge[d], f, n) * Q(umac[d], f, 0) * area;
    }
    const double qvol = 1.0 / (g->dx[0] * g->dx[1] * g->dx[2]);
    int any_convective = 0;
    for (int n = 0; n < ncomp; ++n) if (!iconserv[n]) any_convective = 1;
    for (int n = 0; n < ncomp; ++n)
    _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        /* ComputeDivergence with mult = -1 */
        double upd = -1.0 * qvol * ((A4(&flux[0], i + 1, j, k, n) - A4(&flux[0], i, j, k, n))
                                  + (A4(&flux[1], i, j + 1, k, n) - A4(&flux[1], i, j, k, n))
                                  + (A4(&flux[2], i, j, k + 1, n) - A4(&flux[2], i, j, k, n)));
        if (any_convective && !iconserv[n]) {
            double divum = 1.0 * ((A4(umac[0], i + 1, j, k, 0) - A4(umac[0], i, j, k, 0)) / g->dx[0]
                                + (A4(umac[1], i, j + 1, k, 0) - A4(umac[1], i, j, k, 0)) / g->dx[1]
                                + (A4(umac[2], i, j, k + 1, 0) - A4(umac[2], i, j, k, 0)) / g->dx[2]);
            double qavg = A4(&edge[0], i, j, k, n) + A4(&edge[0], i + 1, j, k, n)
                        + A4(&edge[1], i, j, k, n) + A4(&edge[1], i, j + 1, k, n)
                        + A4(&edge[2], i, j, k, n) + A4(&edge[2], i, j, k + 1, n);
            qavg *= 1.0 / 6.0;
            upd += qavg * divum;
        }
        A4(aofs, i, j, k, acomp + n) = -upd;
    }
    for (int d = 0; d < 3; ++d) {
        if (edge_out && edge_out[d]) memcpy(edge_out[d]->p, edge[d].p, orc_npts(&edge[d]) * ncomp * sizeof(double));
        if (flux_out && flux_out[d]) memcpy(flux_out[d]->p, flux[d].p, orc_npts(&flux[d]) * ncomp * sizeof(double));
        orc_free(&edge[d]); orc_free(&flux[d]);
    }
}

/* NavierStokesBase::ComputeAofs with is_sync = true (Source/NavierStokesBase.cpp:4594-4845 as called from MacProj::mac_sync_compute,
 * Source/MacProj.cpp:700-731): the edge states are traced with the level's u_mac, the fluxes are formed with the correction velocity
 * Ucorr (:4681-4683), the update is the conservative one for every component (:4777: no convective term in a sync) and the result
 * is accumulated: sync -= update with update = -div(F)/vol (:4826-4832). */
void orc_compute_aofs_sync(const orc_geom* g, orc_fab* sync, int acomp, const orc_fab* S, int ncomp,
                           const orc_fab* force, const orc_fab* divu, orc_fab* const umac[3], orc_fab* const ucorr[3], const int* iconserv,
                           double dt, const orc_bcrec* bc, int is_velocity, int use_forces_in_trans, orc_fab* flux_out[3])
{
    orc_fab edge[3], flux[3];
    for (int d = 0; d < 3; ++d) { edge[d] = alloc_faces(g, d, 0, ncomp); flux[d] = alloc_faces(g, d, 0, ncomp); }
    compute_edge_state(g, S, ncomp, force, divu, umac, iconserv, dt, bc, is_velocity, use_forces_in_trans, edge);
    for (int d = 0; d < 3; ++d) {
        const double area = g->dx[(d + 1) % 3] * g->dx[(d + 2) % 3];
        for (int n = 0; n < ncomp; ++n)
        LOOP3(&flux[d], f) *QP(&flux[d], f, n) = Q(&edge[d], f, n) * Q(ucorr[d], f, 0) * area;
    }
    const double qvol = 1.0 / (g->dx[0] * g->dx[1] * g->dx[2]);
    for (int n = 0; n < ncomp; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        const double upd = -1.0 * qvol * ((A4(&flux[0], i + 1, j, k, n) - A4(&flux[0], i, j, k, n))
                                        + (A4(&flux[1], i, j + 1, k, n) - A4(&flux[1], i, j, k, n))
                                        + (A4(&flux[2], i, j, k + 1, n) - A4(&flux[2], i, j, k, n)));
        A4(sync, i, j, k, acomp + n) -= upd;
    }
    for (int d = 0; d < 3; ++d) {
        if (flux_out && flux_out[d]) memcpy(flux_out[d]->p, flux[d].p, orc_npts(&flux[d]) * ncomp * sizeof(double));
        orc_free(&edge[d]); orc_free(&flux[d]);
    }
}
