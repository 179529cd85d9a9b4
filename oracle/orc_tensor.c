/* oracle/orc_tensor.c -- tensor (full viscous stress) operator = 3-component ABec + explicit
 * cross-derivative face fluxes, and its multigrid solve, restated on the CPU (test infrastructure
 * only; PARITY UNPINNED, see orc.h).
 *
 * Follows (upstream AMReX, not in /root/reference): MLTensorOp (setShearViscosity: b_d(comp) =
 * eta*(comp==d ? 4/3 : 1), bulk kappa = 0; mltensor_cross_terms_f{x,y,z}; mltensor_cross_terms).
 * The smoother acts on the ABec part only; the residual includes the cross terms.
 * Reference call sites: Source/Diffusion.cpp:650-957 (diffuse_tensor_velocity) and :1655-1777
 * (getTensorViscTerms).
 */
#include "orc_int.h"

int orc_abec_is_tensor(const orc_abec_level* L) { return L->tensor; }

void orc_tensor_cross_terms_add(const orc_abec_level* L, orc_fab* y, const orc_fab* v)
{
    const orc_geom* g = &L->g;
    const double dxi = 1.0 / g->dx[0], dyi = 1.0 / g->dx[1], dzi = 1.0 / g->dx[2];
    const double twoThirds = 2.0 / 3.0;
    orc_fab fx = orc_alloc(g->n, ORC_FACE[0], 0, 3), fy = orc_alloc(g->n, ORC_FACE[1], 0, 3), fz = orc_alloc(g->n, ORC_FACE[2], 0, 3);
    const orc_fab *etax = &L->b[0], *etay = &L->b[1], *etaz = &L->b[2];
    const double xif = 0.0; /* bulk viscosity kappa = 0 */
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
        double dudy = (A4(v, i, j + 1, k, 0) + A4(v, i - 1, j + 1, k, 0) - A4(v, i, j - 1, k, 0) - A4(v, i - 1, j - 1, k, 0)) * (0.25 * dyi);
        double dvdy = (A4(v, i, j + 1, k, 1) + A4(v, i - 1, j + 1, k, 1) - A4(v, i, j - 1, k, 1) - A4(v, i - 1, j - 1, k, 1)) * (0.25 * dyi);
        double dudz = (A4(v, i, j, k + 1, 0) + A4(v, i - 1, j, k + 1, 0) - A4(v, i, j, k - 1, 0) - A4(v, i - 1, j, k - 1, 0)) * (0.25 * dzi);
        double dwdz = (A4(v, i, j, k + 1, 2) + A4(v, i - 1, j, k + 1, 2) - A4(v, i, j, k - 1, 2) - A4(v, i - 1, j, k - 1, 2)) * (0.25 * dzi);
        double divu = dvdy + dwdz;
        double mun = 0.75 * (A4(etax, i, j, k, 0) - xif);
        double mut = A4(etax, i, j, k, 1);
        A4(&fx, i, j, k, 0) = -mun * (-twoThirds * divu) - xif * divu;
        A4(&fx, i, j, k, 1) = -mut * dudy;
        A4(&fx, i, j, k, 2) = -mut * dudz;
    }
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        double dudx = (A4(v, i + 1, j, k, 0) + A4(v, i + 1, j - 1, k, 0) - A4(v, i - 1, j, k, 0) - A4(v, i - 1, j - 1, k, 0)) * (0.25 * dxi);
        double dvdx = (A4(v, i + 1, j, k, 1) + A4(v, i + 1, j - 1, k, 1) - A4(v, i - 1, j, k, 1) - A4(v, i - 1, j - 1, k, 1)) * (0.25 * dxi);
        double dvdz = (A4(v, i, j, k + 1, 1) + A4(v, i, j - 1, k + 1, 1) - A4(v, i, j, k - 1, 1) - A4(v, i, j - 1, k - 1, 1)) * (0.25 * dzi);
        double dwdz = (A4(v, i, j, k + 1, 2) + A4(v, i, j - 1, k + 1, 2) - A4(v, i, j, k - 1, 2) - A4(v, i, j - 1, k - 1, 2)) * (0.25 * dzi);
        double divu = dudx + dwdz;
        double mun = 0.75 * (A4(etay, i, j, k, 1) - xif);
        double mut = A4(etay, i, j, k, 0);
        A4(&fy, i, j, k, 0) = -mut * dvdx;
        A4(&fy, i, j, k, 1) = -mun * (-twoThirds * divu) - xif * divu;
        A4(&fy, i, j, k, 2) = -mut * dvdz;
    }
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        double dudx = (A4(v, i + 1, j, k, 0) + A4(v, i + 1, j, k - 1, 0) - A4(v, i - 1, j, k, 0) - A4(v, i - 1, j, k - 1, 0)) * (0.25 * dxi);
        double dwdx = (A4(v, i + 1, j, k, 2) + A4(v, i + 1, j, k - 1, 2) - A4(v, i - 1, j, k, 2) - A4(v, i - 1, j, k - 1, 2)) * (0.25 * dxi);
        double dvdy = (A4(v, i, j + 1, k, 1) + A4(v, i, j + 1, k - 1, 1) - A4(v, i, j - 1, k, 1) - A4(v, i, j - 1, k - 1, 1)) * (0.25 * dyi);
        double dwdy = (A4(v, i, j + 1, k, 2) + A4(v, i, j + 1, k - 1, 2) - A4(v, i, j - 1, k, 2) - A4(v, i, j - 1, k - 1, 2)) * (0.25 * dyi);
        double divu = dudx + dvdy;
        double mun = 0.75 * (A4(etaz, i, j, k, 2) - xif);
        double mut = A4(etaz, i, j, k, 0);
        A4(&fz, i, j, k, 0) = -mut * dwdx;
        A4(&fz, i, j, k, 1) = -mut * dwdy;
        A4(&fz, i, j, k, 2) = -mun * (-twoThirds * divu) - xif * divu;
    }
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(y, i, j, k, n) += L->beta * (dxi * (A4(&fx, i + 1, j, k, n) - A4(&fx, i, j, k, n))
                                      + dyi * (A4(&fy, i, j + 1, k, n) - A4(&fy, i, j, k, n))
                                      + dzi * (A4(&fz, i, j, k + 1, n) - A4(&fz, i, j, k, n)));
    orc_free(&fx); orc_free(&fy); orc_free(&fz);
}

static void build_level(const orc_geom* g, orc_abec_level* L, double alpha, double beta, const orc_fab* a, orc_fab* const eta[3])
{
    memset(L, 0, sizeof(*L));
    L->g = *g; L->alpha = alpha; L->beta = beta; L->ncomp = 3; L->tensor = 1;
    if (a) L->a = *a; else L->a.p = NULL;
    for (int d = 0; d < 3; ++d) {
        L->b[d] = orc_alloc(g->n, ORC_FACE[d], 0, 3);
        int hi[3] = {g->n[0] - 1, g->n[1] - 1, g->n[2] - 1}; hi[d] += 1;
        for (int n = 0; n < 3; ++n)
        for (int k = 0; k <= hi[2]; ++k) for (int j = 0; j <= hi[1]; ++j) for (int i = 0; i <= hi[0]; ++i)
            A4(&L->b[d], i, j, k, n) = A4(eta[d], i, j, k, 0) * (n == d ? 4.0 / 3.0 : 1.0);
    }
}

void orc_tensor_apply(const orc_geom* g, orc_fab* y, const orc_fab* u, double alpha, double beta,
                      const orc_fab* a, orc_fab* const eta[3])
{
    orc_abec_level L;
    build_level(g, &L, alpha, beta, a, eta);
    orc_abec_apply(&L, y, u);
    for (int d = 0; d < 3; ++d) orc_free(&L.b[d]);
}

void orc_tensor_solve(const orc_geom* g, orc_fab* u, const orc_fab* rhs, double alpha, double beta,
                      const orc_fab* a, orc_fab* const eta[3], const int lobc[3], const int hibc[3],
                      double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st)
{
    orc_abec_level L;
    build_level(g, &L, alpha, beta, a, eta);
    orc_abec_solve(&L, u, rhs, lobc, hibc, rtol, atol, o, st);
    for (int d = 0; d < 3; ++d) orc_free(&L.b[d]);
}
