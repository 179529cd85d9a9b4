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

/* Edge and corner ghost cells needed by the cross terms (MLTensorOp::applyBCTensor -> mltensor_fill_edges/corners).
 * A ghost cell that is outside the box in two or three directions and, in at least one of them, outside a
 * NON-periodic domain face gets the AVERAGE over those exterior directions d of the one-dimensional BC rule applied
 * along d (Neumann: copy of the neighbour; Dirichlet: the same extrapolation polynomial as on faces, using the ghost
 * cells already filled with one exterior direction less, plus the boundary value stored in that cell of bcval).
 * Cells whose outside directions are all periodic are plain periodic images. */
static void lagr(double xi, const double* x, int N, double* c)
{
    for (int j = 0; j < N; ++j) {
        double num = 1.0, den = 1.0;
        for (int i = 0; i < N; ++i) { if (i == j) continue; num *= xi - x[i]; den *= x[j] - x[i]; }
        c[j] = num / den;
    }
}

void orc_tensor_fill_edges_corners(const orc_abec_level* L, orc_fab* phi, const int lobc[3], const int hibc[3],
                                   int maxorder, int inhomog, const orc_fab* bcval)
{
    const orc_geom* g = &L->g;
    int anywall = 0;
    for (int d = 0; d < 3; ++d) if (!g->periodic[d]) anywall = 1;
    if (!anywall) return;
    for (int nout = 2; nout <= 3; ++nout)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
        const int idx[3] = {i, j, k};
        int out[3], cnt = 0, next = 0;
        for (int d = 0; d < 3; ++d) { out[d] = idx[d] < 0 ? -1 : (idx[d] > g->n[d] - 1 ? 1 : 0); if (out[d]) ++cnt; }
        if (cnt != nout) continue;
        for (int d = 0; d < 3; ++d) if (out[d] && !g->periodic[d]) ++next;
        if (next == 0) continue;
        for (int n = 0; n < L->ncomp; ++n) {
            double sum = 0.0;
            for (int d = 0; d < 3; ++d) {
                if (!out[d] || g->periodic[d]) continue;
                const int bct = out[d] < 0 ? lobc[(L->bc_percomp ? 3 * n : 0) + d] : hibc[(L->bc_percomp ? 3 * n : 0) + d];
                const int s = out[d] < 0 ? 1 : -1;
                double v;
                if (bct == ORC_LO_NEUMANN || bct == ORC_LO_REFLECT_ODD) {
                    int q[3] = {i, j, k}; q[d] += s;
                    v = A4(phi, q[0], q[1], q[2], n);
                    if (bct == ORC_LO_REFLECT_ODD) v = -v;
                } else {
                    const int NX = g->n[d] + 1 < maxorder ? g->n[d] + 1 : maxorder;
                    const double bv = (inhomog && bcval) ? A4(bcval, i, j, k, n) : 0.0;
                    if (NX < 2) v = bv;
                    else {
                        double x[4] = {0.0, 0.5, 1.5, 2.5}, c[4];
                        lagr(-0.5, x, NX, c);
                        double tmp = 0.0;
                        for (int m = 1; m < NX; ++m) { int q[3] = {i, j, k}; q[d] += m * s; tmp += A4(phi, q[0], q[1], q[2], n) * c[m]; }
                        v = tmp + bv * c[0];
                    }
                }
                sum += v;
            }
            A4(phi, i, j, k, n) = sum / (double)next;
        }
    }
}

/* cross-term face fluxes and their divergence for the cells lo..hi; v must hold every cell of the range grown by one */
static void cross_terms_range(const orc_abec_level* L, orc_fab* y, const orc_fab* v, const int lo[3], const int hi[3],
                              orc_fab* fxp, orc_fab* fyp, orc_fab* fzp, orc_fab* const flux_out[3], double flux_scale)
{
    const orc_geom* g = &L->g;
    const double dxi = 1.0 / g->dx[0], dyi = 1.0 / g->dx[1], dzi = 1.0 / g->dx[2];
    const double twoThirds = 2.0 / 3.0;
    orc_fab fx = *fxp, fy = *fyp, fz = *fzp;
    const orc_fab *etax = &L->b[0], *etay = &L->b[1], *etaz = &L->b[2];
    const double xif = 0.0; /* bulk viscosity kappa = 0 */
    _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
    for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0] + 1; ++i) {
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
    _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
    for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1] + 1; ++j) for (int i = lo[0]; i <= hi[0]; ++i) {
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
    _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
    for (int k = lo[2]; k <= hi[2] + 1; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) {
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
    if (y)
    for (int n = 0; n < 3; ++n)
    for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i)
        A4(y, i, j, k, n) += L->beta * (dxi * (A4(&fx, i + 1, j, k, n) - A4(&fx, i, j, k, n))
                                      + dyi * (A4(&fy, i, j + 1, k, n) - A4(&fy, i, j, k, n))
                                      + dzi * (A4(&fz, i, j, k + 1, n) - A4(&fz, i, j, k, n)));
    if (flux_out) {                       /* MLTensorOp::compFlux: the cross-term part of the face fluxes, flux += beta * f */
        orc_fab* fl[3] = {&fx, &fy, &fz};
        for (int d = 0; d < 3; ++d) {
            int h[3] = {hi[0], hi[1], hi[2]}; h[d] += 1;
            for (int n = 0; n < 3; ++n)
            for (int k = lo[2]; k <= h[2]; ++k) for (int j = lo[1]; j <= h[1]; ++j) for (int i = lo[0]; i <= h[0]; ++i)
                A4(flux_out[d], i, j, k, n) += flux_scale * L->beta * A4(fl[d], i, j, k, n);
        }
    }
}

void orc_tensor_cross_terms_add(const orc_abec_level* L, orc_fab* y, const orc_fab* v)
{
    const orc_geom* g = &L->g;
    orc_fab fx = orc_alloc(g->n, ORC_FACE[0], 0, 3), fy = orc_alloc(g->n, ORC_FACE[1], 0, 3), fz = orc_alloc(g->n, ORC_FACE[2], 0, 3);
    const int lo[3] = {0, 0, 0}, hi[3] = {g->n[0] - 1, g->n[1] - 1, g->n[2] - 1};
    cross_terms_range(L, y, v, lo, hi, &fx, &fy, &fz, NULL, 0.0);
    orc_free(&fx); orc_free(&fy); orc_free(&fz);
}

/* the same on a level that does not cover the domain: box by box, on a copy of the box grown by one cell whose ghost cells hold what
 * the box sees there (orc_cf_box_value).  flux_out (may be NULL): the cross-term fluxes are ADDED to it, scaled, on the faces of every box */
double orc_cf_box_value(const orc_abec_level* L, const orc_fab* x, const int* bx, int i, int j, int k, int n);   /* orc_abec.c */
static void cross_terms_cf(const orc_abec_level* L, orc_fab* y, const orc_fab* x, orc_fab* const flux_out[3], double flux_scale)
{
    const orc_geom* g = &L->g;
    orc_fab fx = orc_alloc(g->n, ORC_FACE[0], 0, 3), fy = orc_alloc(g->n, ORC_FACE[1], 0, 3), fz = orc_alloc(g->n, ORC_FACE[2], 0, 3);
    for (int b = 0; b < L->nbox; ++b) {
        const int* bx = L->boxes + 6 * b;
        const int bn[3] = {bx[3] - bx[0] + 1, bx[4] - bx[1] + 1, bx[5] - bx[2] + 1};
        orc_fab vl = orc_alloc(bn, ORC_CELL, 1, 3);
        for (int d = 0; d < 3; ++d) { vl.lo[d] += bx[d]; vl.hi[d] += bx[d]; }
        for (int n = 0; n < 3; ++n)
        for (int k = bx[2] - 1; k <= bx[5] + 1; ++k) for (int j = bx[1] - 1; j <= bx[4] + 1; ++j) for (int i = bx[0] - 1; i <= bx[3] + 1; ++i)
            A4(&vl, i, j, k, n) = orc_cf_box_value(L, x, bx, i, j, k, n);
        const int lo[3] = {bx[0], bx[1], bx[2]}, hi[3] = {bx[3], bx[4], bx[5]};
        /* a face shared by two boxes gets the same flux from both (both see the same values there): adding it twice must be avoided */
        if (flux_out) {
            orc_fab t[3]; orc_fab* tp[3];
            for (int d = 0; d < 3; ++d) { t[d] = orc_alloc(g->n, ORC_FACE[d], 0, 3); tp[d] = &t[d]; }
            cross_terms_range(L, y, &vl, lo, hi, &fx, &fy, &fz, tp, flux_scale);
            for (int d = 0; d < 3; ++d) {
                int h[3] = {hi[0], hi[1], hi[2]}; h[d] += 1;
                for (int n = 0; n < 3; ++n)
                for (int k = lo[2]; k <= h[2]; ++k) for (int j = lo[1]; j <= h[1]; ++j) for (int i = lo[0]; i <= h[0]; ++i) {
                    const int f[3] = {i, j, k};
                    /* the low box of a shared face writes it; faces on the box's low side that belong to another box of the level are skipped */
                    if (f[d] == lo[d]) {
                        int c[3] = {i, j, k}; c[d] -= 1;
                        int owned = 0;
                        for (int b2 = 0; b2 < L->nbox && !owned; ++b2) {
                            if (b2 == b) continue;
                            const int* o = L->boxes + 6 * b2;
                            if (c[0] >= o[0] && c[0] <= o[3] && c[1] >= o[1] && c[1] <= o[4] && c[2] >= o[2] && c[2] <= o[5]) owned = 1;
                        }
                        if (owned) continue;
                    }
                    A4(flux_out[d], i, j, k, n) += A4(&t[d], i, j, k, n);
                }
                orc_free(&t[d]);
            }
        } else cross_terms_range(L, y, &vl, lo, hi, &fx, &fy, &fz, NULL, 0.0);
        orc_free(&vl);
    }
    orc_free(&fx); orc_free(&fy); orc_free(&fz);
}
void orc_tensor_cross_terms_add_cf(const orc_abec_level* L, orc_fab* y, const orc_fab* x) { cross_terms_cf(L, y, x, NULL, 0.0); }

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

/* MLMG::apply with setLevelBC(u): the ghost cells of u hold the boundary data on entry (Diffusion::getTensorViscTerms,
 * reference Source/Diffusion.cpp:1655-1777); on exit they hold the operator's ghost values (face, edge, corner fill) */
void orc_tensor_apply_bcn(const orc_geom* g, orc_fab* y, orc_fab* u, double alpha, double beta, const orc_fab* a,
                          orc_fab* const eta[3], const int* lobc, const int* hibc, int maxorder)
{
    orc_abec_level L;
    build_level(g, &L, alpha, beta, a, eta);
    L.bc_percomp = 1;
    orc_fab bcval = orc_alloc(g->n, ORC_CELL, 1, 3);
    orc_copy_all(&bcval, u);
    orc_abec_applybc(&L, u, lobc, hibc, maxorder, 1, &bcval);
    orc_abec_apply(&L, y, u);
    orc_free(&bcval);
    for (int d = 0; d < 3; ++d) orc_free(&L.b[d]);
}

/* lobc/hibc hold 9 codes [n*3+d]: one BC triple per velocity component (Diffusion::setDomainBC per component,
 * reference Source/Diffusion.cpp:724-731 and 1939-2020: slip walls are Dirichlet for the normal and Neumann for the
 * tangential components) */
void orc_tensor_solve_bcn(const orc_geom* g, orc_fab* u, const orc_fab* rhs, double alpha, double beta,
                          const orc_fab* a, orc_fab* const eta[3], const int* lobc, const int* hibc,
                          double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st)
{
    orc_abec_level L;
    build_level(g, &L, alpha, beta, a, eta);
    L.bc_percomp = 1;
    orc_abec_solve(&L, u, rhs, lobc, hibc, rtol, atol, o, st);
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

/* ---- tensor operator on an AMR level that does not cover the domain (Diffusion::getTensorViscTerms / diffuse_tensor_velocity /
 * diffuse_tensor_Vsync at level > 0: tensorop.setCoarseFineBC(&crsedata or nullptr, ratio), Source/Diffusion.cpp:733-744, 876-887,
 * 1096-1099, 1725-1736).  Face ghost cells at coarse/fine faces: the MLCellLinOp formula per component (orc_abec.c).  Edge / corner
 * coarse-fine ghost cells (needed by the cross terms only): the coarse velocity interpolated quadratically to the cell centre
 * (orc_cf_box_value), frozen during the solve, zero in the correction form -- upstream's MLTensorOp::applyBCTensor is not available to follow (PARITY UNPINNED). */
void orc_cf_set_edgeval(const orc_fab* e, int ratio);
void orc_cf_set_bcval(const orc_fab* b, int inhomog, int maxorder);
void orc_cf_interp_bndry(const orc_abec_level* L, int ratio, const orc_fab* cphi, orc_fab* bcval);
void orc_abec_solve_cf(const orc_abec_level* L, orc_fab* phi, const orc_fab* rhs, const int lobc[3], const int hibc[3],
                       const orc_fab* cf_bcval, double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st);
typedef struct tensor_cf { orc_abec_level L; orc_fab bcv; } tensor_cf;
static void tcf_begin(tensor_cf* T, const orc_geom* g, int nbox, const int* boxes, int ratio, double alpha, double beta, const orc_fab* a,
                      orc_fab* const eta[3], const orc_fab* cvel)
{
    build_level(g, &T->L, alpha, beta, a, eta);
    T->L.bc_percomp = 1;
    T->L.nbox = nbox; T->L.boxes = boxes;
    for (int d = 0; d < 3; ++d) T->L.cf_loc[d] = 0.5 * ratio * g->dx[d];
    T->bcv = orc_alloc(g->n, ORC_CELL, 1, 9);
    orc_setval(&T->bcv, 0.0);
    if (cvel) orc_cf_interp_bndry(&T->L, ratio, cvel, &T->bcv);
    orc_cf_set_edgeval(cvel, ratio);
}
static void tcf_end(tensor_cf* T)
{
    orc_cf_set_edgeval(NULL, 2); orc_cf_set_bcval(NULL, 0, 2);
    orc_free(&T->bcv);
    for (int d = 0; d < 3; ++d) orc_free(&T->L.b[d]);
}
/* y = (alpha a - beta div tau) u on the cells of the level; the ghost cells of u outside the physical domain hold the boundary data on entry */
void orc_tensor_apply_cf(const orc_geom* g, int nbox, const int* boxes, int ratio, orc_fab* y, orc_fab* u, double alpha, double beta,
                         const orc_fab* a, orc_fab* const eta[3], const int* lobc, const int* hibc, int maxorder, const orc_fab* cvel)
{
    tensor_cf T;
    tcf_begin(&T, g, nbox, boxes, ratio, alpha, beta, a, eta, cvel);
    orc_fab bcval = orc_alloc(g->n, ORC_CELL, 1, 3);
    orc_copy_all(&bcval, u);
    orc_cf_set_bcval(&T.bcv, 1, maxorder);
    orc_abec_applybc(&T.L, u, lobc, hibc, maxorder, 1, &bcval);
    orc_abec_apply(&T.L, y, u);
    orc_free(&bcval);
    tcf_end(&T);
}
void orc_tensor_solve_cf(const orc_geom* g, int nbox, const int* boxes, int ratio, orc_fab* u, const orc_fab* rhs, double alpha, double beta,
                         const orc_fab* a, orc_fab* const eta[3], const int* lobc, const int* hibc, const orc_fab* cvel,
                         double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st)
{
    tensor_cf T;
    tcf_begin(&T, g, nbox, boxes, ratio, alpha, beta, a, eta, cvel);
    orc_abec_solve_cf(&T.L, u, rhs, lobc, hibc, &T.bcv, rtol, atol, o, st);
    tcf_end(&T);
}

/* Diffusion::computeExtensiveFluxes on the tensor operator (Source/Diffusion.cpp:1463-1537; MLTensorOp::compFlux = ABec flux + cross
 * terms, without the b scalar): flux_d(n) = or += fac * area_d * ( -eta_d (4/3 if n == d) du_n/dx_d + cross_d(n) ) on every face of
 * the level.  u: as the operator left it (ghost cells outside the physical domain filled by applyBC).  nbox > 0: level that does not
 * cover the domain, cvel = the coarse velocity of the coarse/fine data (NULL: homogeneous). */
void orc_abec_extensive_flux(const orc_abec_level* L, orc_fab* flux[3], const orc_fab* phi, double fac, int add);
void orc_tensor_extensive_flux(const orc_geom* g, int nbox, const int* boxes, int ratio, orc_fab* flux[3], const orc_fab* u, orc_fab* const eta[3],
                               double fac, int add, const orc_fab* cvel, int maxorder)
{
    tensor_cf T;
    tcf_begin(&T, g, nbox, boxes, ratio, 0.0, 1.0, NULL, eta, cvel);
    if (nbox == 0) orc_cf_set_edgeval(NULL, 2);
    orc_cf_set_bcval(nbox > 0 ? &T.bcv : NULL, 1, maxorder);
    orc_abec_extensive_flux(&T.L, flux, u, fac, add);              /* b = eta (4/3 on the normal component): the ABec part */
    orc_fab t[3]; orc_fab* tp[3];
    for (int d = 0; d < 3; ++d) { t[d] = orc_alloc(g->n, ORC_FACE[d], 0, 3); orc_setval(&t[d], 0.0); tp[d] = &t[d]; }
    if (nbox > 0) cross_terms_cf(&T.L, NULL, u, tp, 1.0);          /* beta = 1: t += the cross-term face fluxes */
    else {
        orc_fab fx = orc_alloc(g->n, ORC_FACE[0], 0, 3), fy = orc_alloc(g->n, ORC_FACE[1], 0, 3), fz = orc_alloc(g->n, ORC_FACE[2], 0, 3);
        const int lo[3] = {0, 0, 0}, hi[3] = {g->n[0] - 1, g->n[1] - 1, g->n[2] - 1};
        cross_terms_range(&T.L, NULL, u, lo, hi, &fx, &fy, &fz, tp, 1.0);
        orc_free(&fx); orc_free(&fy); orc_free(&fz);
    }
    for (int d = 0; d < 3; ++d) {
        const double sc = fac * g->dx[(d + 1) % 3] * g->dx[(d + 2) % 3];
        const size_t N = orc_npts(&t[d]) * 3;
        for (size_t q = 0; q < N; ++q) flux[d]->p[q] += sc * t[d].p[q];
        orc_free(&t[d]);
    }
    tcf_end(&T);
}
