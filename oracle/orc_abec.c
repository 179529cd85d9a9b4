/* oracle/orc_abec.c -- cell-centred (alpha*a - beta div b grad) operator, geometric multigrid
 * and the MAC projection, restated on the CPU (test infrastructure only; PARITY UNPINNED, see orc.h).
 *
 * Follows (upstream, not in /root/reference): AMReX MLABecLaplacian (mlabeclap_adotx, abec_gsrb,
 * mlabeclap_flux), MLCellLinOp::applyBC (mllinop_apply_bc_x, poly_interp_coeff), MLMG::solve /
 * oneIter / mgVcycle / actualBottomSolve, MLCGSolver::solve_bicgstab, amrex_avgdown(_faces),
 * Hydro::MacProjector::project.  Reference call sites: Source/MacProj.cpp:1084-1184 (coefficients,
 * BCs, max_order=4, tolerances), Source/Diffusion.cpp:327-345.
 */
#include "orc_int.h"

/* threads used by the OpenMP-parallel colour loops of the smoothers (default 1: the oracle is a deterministic checker first;
 * a box whose CPU quota is smaller than its core count would crawl with one spinning thread per core) */
int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; }

void orc_mg_default_opts(orc_mg_opts* o)
{
    o->nu1 = 2; o->nu2 = 2; o->nuf = 8; o->nub = 0;
    o->max_iters = 200; o->bottom_maxiter = 200; o->bottom_reltol = 1.e-4;
    o->omega = 1.15; o->maxorder = 3; o->max_coarsening_level = 30; o->min_width = 2;
    o->nodal_sweeps = 4; o->nodal_smoother = 0; o->verbose = 0; o->bottom_smoother_only = 0;
    o->fixed_iters = 0;
}

/* ---------------------------------------------------------------- operator ----- */
void orc_tensor_cross_terms_add(const orc_abec_level* L, orc_fab* y, const orc_fab* x); /* orc_tensor.c */
int orc_abec_is_tensor(const orc_abec_level* L);                                          /* orc_tensor.c */
void orc_tensor_fill_edges_corners(const orc_abec_level* L, orc_fab* phi, const int lobc[3], const int hibc[3],
                                   int maxorder, int inhomog, const orc_fab* bcval);    /* orc_tensor.c */

static void cf_apply(const orc_abec_level* L, orc_fab* y, const orc_fab* x);
void orc_tensor_cross_terms_add_cf(const orc_abec_level* L, orc_fab* y, const orc_fab* x);   /* orc_tensor.c */
void orc_abec_apply(const orc_abec_level* L, orc_fab* y, const orc_fab* x)
{
    if (L->nbox > 0) { cf_apply(L, y, x); if (orc_abec_is_tensor(L)) orc_tensor_cross_terms_add_cf(L, y, x); return; }
    const orc_geom* g = &L->g;
    const double dhx = L->beta / (g->dx[0] * g->dx[0]);
    const double dhy = L->beta / (g->dx[1] * g->dx[1]);
    const double dhz = L->beta / (g->dx[2] * g->dx[2]);
    const orc_fab *bX = &L->b[0], *bY = &L->b[1], *bZ = &L->b[2];
    for (int n = 0; n < L->ncomp; ++n)
    for (int k = 0; k < g->n[2]; ++k)
    for (int j = 0; j < g->n[1]; ++j)
    for (int i = 0; i < g->n[0]; ++i) {
        double ax = (L->alpha != 0.0 && L->a.p) ? L->alpha * A4(&L->a, i, j, k, 0) * A4(x, i, j, k, n) : 0.0;
        A4(y, i, j, k, n) = ax
            - dhx * (A4(bX, i + 1, j, k, n) * (A4(x, i + 1, j, k, n) - A4(x, i, j, k, n))
                   - A4(bX, i, j, k, n) * (A4(x, i, j, k, n) - A4(x, i - 1, j, k, n)))
            - dhy * (A4(bY, i, j + 1, k, n) * (A4(x, i, j + 1, k, n) - A4(x, i, j, k, n))
                   - A4(bY, i, j, k, n) * (A4(x, i, j, k, n) - A4(x, i, j - 1, k, n)))
            - dhz * (A4(bZ, i, j, k + 1, n) * (A4(x, i, j, k + 1, n) - A4(x, i, j, k, n))
                   - A4(bZ, i, j, k, n) * (A4(x, i, j, k, n) - A4(x, i, j, k - 1, n)));
    }
    if (orc_abec_is_tensor(L)) orc_tensor_cross_terms_add(L, y, x);
}

/* Lagrange weights c[j] = prod_{i!=j} (xi - x[i])/(x[j]-x[i])  (amrex poly_interp_coeff) */
static void poly_interp_coeff(double xi, const double* x, int N, double* c)
{
    for (int j = 0; j < N; ++j) {
        double num = 1.0, den = 1.0;
        for (int i = 0; i < N; ++i) {
            if (i == j) continue;
            num *= xi - x[i];
            den *= x[j] - x[i];
        }
        c[j] = num / den;
    }
}

/* coefficient of the first interior cell in the ghost-cell formula (mllinop_comp_interp_coef0) */
static double bc_coef0(int bct, int blen, int maxorder)
{
    if (bct == ORC_LO_NEUMANN) return 1.0;
    if (bct == ORC_LO_REFLECT_ODD) return -1.0;
    if (bct == ORC_LO_DIRICHLET) {
        int NX = blen + 1 < maxorder ? blen + 1 : maxorder;
        if (NX < 2) return 0.0;
        double x[4] = {0.0, 0.5, 1.5, 2.5}, c[4];
        poly_interp_coeff(-0.5, x, NX, c);
        return c[1];
    }
    return 0.0;
}

#define BCOFF(L, n) ((L)->bc_percomp ? 3 * (n) : 0)

/* ------------------------------------------------------------- coarse/fine faces (level with nbox > 0) ----
 * The level's cells are the union of its boxes.  The operator is evaluated box by box on a private copy of the box grown by
 * one cell -- exactly how MLCellLinOp sees its data: ghost cells covered by another box (or a periodic image) carry that
 * box's value, ghost cells outside the physical domain carry the domain-BC ghost value of the global array, and the remaining
 * ghost cells are coarse/fine ghost cells filled by the Dirichlet formula of mllinop_apply_bc with the boundary value
 * cf_loc[d] behind the face: ghost = c[0]*bcval + sum_m c[m]*phi(m-th cell inside), Lagrange weights through
 * x = {-cf_loc/dx, 0.5, 1.5, 2.5}, order NX = min(box length + 1, maxorder). */
static const orc_fab* g_cf_bcval = NULL;     /* coarse/fine Dirichlet data of the solve in progress */
static const orc_fab* g_cf_edgeval = NULL;   /* tensor operator: the coarse data (cell centred, >= 1 filled ghost cell) behind the edge / corner coarse-fine ghost cells */
static int g_cf_edge_ratio = 2;
/* domain boundary conditions of the operator in use (set by orc_abec_applybc): needed where a coarse/fine ghost column meets a wall */
static const int *g_bc_lobc = NULL, *g_bc_hibc = NULL;
static const orc_fab* g_bc_bcval = NULL;
static int g_cf_inhomog = 0;                 /* set by orc_abec_applybc: the next apply uses the data (1) or zero (0) */
static int g_cf_maxorder = 2;

static int box_of(const orc_abec_level* L, int i, int j, int k)
{
    /* index (1-based) of the box holding cell (i,j,k) after periodic wrap; 0: inside the domain, not covered; -1: outside */
    int c[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) {
        if (c[d] < 0 || c[d] >= L->g.n[d]) {
            if (!L->g.periodic[d]) return -1;
            c[d] = (c[d] % L->g.n[d] + L->g.n[d]) % L->g.n[d];
        }
    }
    for (int b = 0; b < L->nbox; ++b) {
        const int* bx = L->boxes + 6 * b;
        if (c[0] >= bx[0] && c[0] <= bx[3] && c[1] >= bx[1] && c[1] <= bx[4] && c[2] >= bx[2] && c[2] <= bx[5]) return b + 1;
    }
    return 0;
}
static inline double wrapped(const orc_abec_level* L, const orc_fab* x, int i, int j, int k, int n)
{
    int c[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) if (L->g.periodic[d]) c[d] = (c[d] % L->g.n[d] + L->g.n[d]) % L->g.n[d];
    return A4(x, c[0], c[1], c[2], n);
}
static void cf_coefs(const orc_abec_level* L, int d, int blen, int maxorder, double c[4], int* NXo)
{
    int NX = blen + 1 < maxorder ? blen + 1 : maxorder;
    double x[4] = {-L->cf_loc[d] / L->g.dx[d], 0.5, 1.5, 2.5};
    c[0] = c[1] = c[2] = c[3] = 0.0;
    if (NX >= 2) poly_interp_coeff(-0.5, x, NX, c);
    *NXo = NX;
}
/* value of x in the cell next to cell (i,j,k) of box bx in direction d, side s = -1 / +1, as the box sees it */
static double box_nbr(const orc_abec_level* L, const orc_fab* x, const int* bx, int i, int j, int k, int n, int d, int s, int* is_cf)
{
    int q[3] = {i, j, k};
    q[d] += s;
    *is_cf = 0;
    if (q[d] >= bx[d] && q[d] <= bx[3 + d]) return A4(x, q[0], q[1], q[2], n);          /* inside the box */
    const int w = box_of(L, q[0], q[1], q[2]);
    if (w < 0) return A4(x, q[0], q[1], q[2], n);                                        /* physical boundary: stored ghost value */
    if (w > 0) return wrapped(L, x, q[0], q[1], q[2], n);                                /* another box / periodic image */
    *is_cf = 1;
    double c[4]; int NX;
    cf_coefs(L, d, bx[3 + d] - bx[d] + 1, g_cf_maxorder, c, &NX);
    const double bv = (g_cf_inhomog && g_cf_bcval) ? A4(g_cf_bcval, q[0], q[1], q[2], n * 3 + d) : 0.0;
    if (NX < 2) return bv;
    double v = bv * c[0];
    for (int m = 1; m < NX; ++m) {
        int r[3] = {i, j, k};
        r[d] -= s * (m - 1);
        v += c[m] * A4(x, r[0], r[1], r[2], n);
    }
    return v;
}
/* value of x in cell (i,j,k) of the grown box bx (one ghost layer) as the box sees it: valid data of the level (own box, other boxes,
 * periodic images), the stored ghost value outside the physical domain, the coarse/fine face formula next to a face of the box, and
 * in the edge / corner coarse-fine ghost cells (tensor cross terms only) the coarse data interpolated to the cell centre -- frozen
 * during the solve, zero in the homogeneous (correction) form of the operator */
double orc_cf_box_value(const orc_abec_level* L, const orc_fab* x, const int* bx, int i, int j, int k, int n)
{
    const int q[3] = {i, j, k};
    int nout = 0, d = -1, s = 0;
    for (int e = 0; e < 3; ++e) {
        if (q[e] < bx[e]) { ++nout; d = e; s = -1; }
        else if (q[e] > bx[3 + e]) { ++nout; d = e; s = 1; }
    }
    if (nout == 0) return A4(x, i, j, k, n);
    const int w = box_of(L, i, j, k);
    if (w < 0) {
        /* outside the physical domain.  If the cell's projection into the domain is a cell of the level, the stored ghost value (formed by
         * orc_abec_applybc from that column of level cells) is what the box sees.  Otherwise the column next to the wall consists of
         * coarse/fine ghost cells of this box: the one-dimensional boundary rule is applied to what the box sees there, averaged over the
         * exterior directions (MLTensorOp::applyBCTensor's edge / corner fill, as orc_tensor_fill_edges_corners) */
        int qc[3] = {i, j, k};
        for (int e = 0; e < 3; ++e) if (!L->g.periodic[e]) { if (qc[e] < 0) qc[e] = 0; if (qc[e] > L->g.n[e] - 1) qc[e] = L->g.n[e] - 1; }
        const int inbox = qc[0] >= bx[0] && qc[0] <= bx[3] && qc[1] >= bx[1] && qc[1] <= bx[4] && qc[2] >= bx[2] && qc[2] <= bx[5];
        if (inbox || box_of(L, qc[0], qc[1], qc[2]) != 0 || !g_bc_lobc) return A4(x, i, j, k, n);
        double sum = 0.0; int cnt = 0;
        for (int e = 0; e < 3; ++e) {
            if (L->g.periodic[e] || (q[e] >= 0 && q[e] <= L->g.n[e] - 1)) continue;
            const int sg = q[e] < 0 ? 1 : -1;
            const int bct = q[e] < 0 ? g_bc_lobc[BCOFF(L, n) + e] : g_bc_hibc[BCOFF(L, n) + e];
            double v;
            if (bct == ORC_LO_NEUMANN || bct == ORC_LO_REFLECT_ODD) {
                int r[3] = {i, j, k}; r[e] += sg;
                v = orc_cf_box_value(L, x, bx, r[0], r[1], r[2], n);
                if (bct == ORC_LO_REFLECT_ODD) v = -v;
            } else {
                const int NX = L->g.n[e] + 1 < g_cf_maxorder ? L->g.n[e] + 1 : g_cf_maxorder;
                const double bv = (g_cf_inhomog && g_bc_bcval) ? A4(g_bc_bcval, i, j, k, n) : 0.0;
                if (NX < 2) v = bv;
                else {
                    double xs[4] = {0.0, 0.5, 1.5, 2.5}, c[4] = {0, 0, 0, 0};
                    poly_interp_coeff(-0.5, xs, NX, c);
                    v = bv * c[0];
                    for (int m = 1; m < NX; ++m) { int r[3] = {i, j, k}; r[e] += m * sg; v += c[m] * orc_cf_box_value(L, x, bx, r[0], r[1], r[2], n); }
                }
            }
            sum += v; ++cnt;
        }
        return sum / (double)cnt;
    }
    if (w > 0) return wrapped(L, x, i, j, k, n);
    if (nout == 1) {
        int r[3] = {i, j, k}, cf;
        r[d] -= s;                                  /* the cell of the box next to the face */
        return box_nbr(L, x, bx, r[0], r[1], r[2], n, d, s, &cf);
    }
    if (!(g_cf_inhomog && g_cf_edgeval)) return 0.0;
    /* coarse data interpolated to the cell centre, quadratically in every direction: centred stencil where the cell lies inside the
     * box's index range, one-sided towards the box where it lies outside */
    const int r = g_cf_edge_ratio;
    int c[3], o[3][3];
    double wq[3][3];
    for (int e = 0; e < 3; ++e) {
        c[e] = q[e] >= 0 ? q[e] / r : -((-q[e] + r - 1) / r);
        const double off = (q[e] - c[e] * r + 0.5) / r - 0.5;
        if (q[e] < bx[e] || q[e] > bx[3 + e]) {
            const int sg = q[e] < bx[e] ? 1 : -1;
            const double u = off * sg;
            o[e][0] = 0; o[e][1] = sg; o[e][2] = 2 * sg;
            wq[e][0] = 0.5 * (u - 1.0) * (u - 2.0); wq[e][1] = -u * (u - 2.0); wq[e][2] = 0.5 * u * (u - 1.0);
        } else if (!L->g.periodic[e] && c[e] - 1 < 0) {             /* next to a wall: one-sided, no coarse cell outside the physical domain */
            o[e][0] = 0; o[e][1] = 1; o[e][2] = 2;
            wq[e][0] = 0.5 * (off - 1.0) * (off - 2.0); wq[e][1] = -off * (off - 2.0); wq[e][2] = 0.5 * off * (off - 1.0);
        } else if (!L->g.periodic[e] && c[e] + 1 > L->g.n[e] / r - 1) {
            o[e][0] = -2; o[e][1] = -1; o[e][2] = 0;
            wq[e][0] = 0.5 * off * (off + 1.0); wq[e][1] = -off * (off + 2.0); wq[e][2] = 0.5 * (off + 1.0) * (off + 2.0);
        } else {
            o[e][0] = -1; o[e][1] = 0; o[e][2] = 1;
            wq[e][0] = 0.5 * off * (off - 1.0); wq[e][1] = 1.0 - off * off; wq[e][2] = 0.5 * off * (off + 1.0);
        }
    }
    const int cn[3] = {L->g.n[0] / r, L->g.n[1] / r, L->g.n[2] / r};
    double v = 0.0;
    for (int cz = 0; cz < 3; ++cz) for (int cy = 0; cy < 3; ++cy) for (int cx = 0; cx < 3; ++cx) {
        int z[3] = {c[0] + o[0][cx], c[1] + o[1][cy], c[2] + o[2][cz]};
        for (int e = 0; e < 3; ++e) if (L->g.periodic[e]) z[e] = (z[e] % cn[e] + cn[e]) % cn[e];
        v += wq[0][cx] * wq[1][cy] * wq[2][cz] * A4(g_cf_edgeval, z[0], z[1], z[2], n);
    }
    return v;
}
void orc_cf_set_edgeval(const orc_fab* e, int ratio) { g_cf_edgeval = e; g_cf_edge_ratio = ratio; }
void orc_cf_set_bcval(const orc_fab* b, int inhomog, int maxorder) { g_cf_bcval = b; g_cf_inhomog = inhomog; g_cf_maxorder = maxorder; }
static void cf_zero_uncovered(const orc_abec_level* L, orc_fab* y)
{
    const orc_geom* g = &L->g;
    for (int n = 0; n < y->nc; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        if (box_of(L, i, j, k) == 0) A4(y, i, j, k, n) = 0.0;
}
static void cf_apply(const orc_abec_level* L, orc_fab* y, const orc_fab* x)
{
    const orc_geom* g = &L->g;
    const double dh[3] = {L->beta / (g->dx[0] * g->dx[0]), L->beta / (g->dx[1] * g->dx[1]), L->beta / (g->dx[2] * g->dx[2])};
    cf_zero_uncovered(L, y);
    for (int b = 0; b < L->nbox; ++b) {
        const int* bx = L->boxes + 6 * b;
        for (int n = 0; n < L->ncomp; ++n)
        for (int k = bx[2]; k <= bx[5]; ++k) for (int j = bx[1]; j <= bx[4]; ++j) for (int i = bx[0]; i <= bx[3]; ++i) {
            const double xc = A4(x, i, j, k, n);
            double v = (L->alpha != 0.0 && L->a.p) ? L->alpha * A4(&L->a, i, j, k, 0) * xc : 0.0;
            for (int d = 0; d < 3; ++d) {
                int f[3] = {i, j, k}, cf;
                const double blo = A4(&L->b[d], f[0], f[1], f[2], n);
                f[d] += 1;
                const double bhi = A4(&L->b[d], f[0], f[1], f[2], n);
                const double xm = box_nbr(L, x, bx, i, j, k, n, d, -1, &cf), xp = box_nbr(L, x, bx, i, j, k, n, d, +1, &cf);
                v -= dh[d] * (bhi * (xp - xc) - blo * (xc - xm));
            }
            A4(y, i, j, k, n) = v;
        }
    }
}
static void cf_gsrb(const orc_abec_level* L, orc_fab* phi, const orc_fab* rhs, int redblack, double omega,
                    const int lobc[3], const int hibc[3], int maxorder)
{
    const orc_geom* g = &L->g;
    const double dh[3] = {L->beta / (g->dx[0] * g->dx[0]), L->beta / (g->dx[1] * g->dx[1]), L->beta / (g->dx[2] * g->dx[2])};
    for (int b = 0; b < L->nbox; ++b) {
        const int* bx = L->boxes + 6 * b;
        /* ghost cells of the box as MLCellLinOp::applyBC leaves them in front of the colour pass: formed from the state BEFORE
         * the pass (the coarse/fine formula reaches up to three cells into the box, i.e. cells of both colours) */
        const int gn[3] = {bx[3] - bx[0] + 3, bx[4] - bx[1] + 3, bx[5] - bx[2] + 3};
        double* gh = (double*)malloc(sizeof(double) * (size_t)gn[0] * gn[1] * gn[2] * L->ncomp);
        unsigned char* gcf = (unsigned char*)calloc((size_t)gn[0] * gn[1] * gn[2], 1);
#define GH(i, j, k, n) gh[((size_t)(n) * gn[2] + ((k) - bx[2] + 1)) * gn[1] * gn[0] + (size_t)((j) - bx[1] + 1) * gn[0] + ((i) - bx[0] + 1)]
#define GCF(i, j, k) gcf[((size_t)((k) - bx[2] + 1) * gn[1] + ((j) - bx[1] + 1)) * gn[0] + ((i) - bx[0] + 1)]
        for (int n = 0; n < L->ncomp; ++n)
        for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
            int lo[3] = {bx[0], bx[1], bx[2]}, hi[3] = {bx[3], bx[4], bx[5]};
            lo[d] = hi[d] = side == 0 ? bx[d] : bx[3 + d];
            for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) {
                int q[3] = {i, j, k}, cf;
                q[d] += side == 0 ? -1 : 1;
                GH(q[0], q[1], q[2], n) = box_nbr(L, phi, bx, i, j, k, n, d, side == 0 ? -1 : 1, &cf);
                GCF(q[0], q[1], q[2]) = (unsigned char)cf;
            }
        }
        for (int n = 0; n < L->ncomp; ++n)
        for (int k = bx[2]; k <= bx[5]; ++k) for (int j = bx[1]; j <= bx[4]; ++j) for (int i = bx[0]; i <= bx[3]; ++i) {
            if ((i + j + k + redblack) % 2 != 0) continue;
            const int idx[3] = {i, j, k};
            const double aa = (L->alpha != 0.0 && L->a.p) ? L->alpha * A4(&L->a, i, j, k, 0) : 0.0;
            double gamma = aa, corr = 0.0, rho = 0.0;
            for (int d = 0; d < 3; ++d) {
                int f[3] = {i, j, k}, m[3] = {i, j, k}, p[3] = {i, j, k};
                const double blo = A4(&L->b[d], f[0], f[1], f[2], n);
                f[d] += 1;
                const double bhi = A4(&L->b[d], f[0], f[1], f[2], n);
                m[d] -= 1; p[d] += 1;
                const int olo = idx[d] == bx[d], ohi = idx[d] == bx[3 + d];
                const double xm = olo ? GH(m[0], m[1], m[2], n) : A4(phi, m[0], m[1], m[2], n);
                const double xp = ohi ? GH(p[0], p[1], p[2], n) : A4(phi, p[0], p[1], p[2], n);
                const int cfl = olo && GCF(m[0], m[1], m[2]), cfh = ohi && GCF(p[0], p[1], p[2]);
                /* coefficient of the first interior cell in the ghost formula (mllinop_comp_interp_coef0) */
                double cl = 0.0, ch = 0.0, c[4]; int NX;
                if (cfl || cfh) cf_coefs(L, d, bx[3 + d] - bx[d] + 1, maxorder, c, &NX);
                if (cfl) cl = c[1];
                else if (!g->periodic[d] && idx[d] == 0) cl = bc_coef0(lobc[BCOFF(L, n) + d], g->n[d], maxorder);
                if (cfh) ch = c[1];
                else if (!g->periodic[d] && idx[d] == g->n[d] - 1) ch = bc_coef0(hibc[BCOFF(L, n) + d], g->n[d], maxorder);
                gamma += dh[d] * (blo + bhi);
                corr += dh[d] * (blo * cl + bhi * ch);
                rho += dh[d] * (blo * xm + bhi * xp);
            }
            const double res = A4(rhs, i, j, k, n) - (gamma * A4(phi, i, j, k, n) - rho);
            A4(phi, i, j, k, n) = A4(phi, i, j, k, n) + omega / (gamma - corr) * res;
        }
#undef GH
#undef GCF
        free(gh); free(gcf);
    }
}

void orc_abec_applybc(const orc_abec_level* L, orc_fab* phi, const int lobc[3], const int hibc[3],
                      int maxorder, int inhomog, const orc_fab* bcval)
{
    const orc_geom* g = &L->g;
    g_cf_inhomog = inhomog; g_cf_maxorder = maxorder;      /* coarse/fine ghost values are formed on the fly (box_nbr) */
    {   /* kept as copies: the flux evaluation that follows an apply / solve (orc_*_extensive_flux) still needs them */
        static int lo_copy[9], hi_copy[9];
        static orc_fab bcv_copy = {NULL, {0, 0, 0}, {0, 0, 0}, 0};
        const int nb = L->bc_percomp ? 3 * L->ncomp : 3;
        for (int q = 0; q < nb && q < 9; ++q) { lo_copy[q] = lobc[q]; hi_copy[q] = hibc[q]; }
        g_bc_lobc = lo_copy; g_bc_hibc = hi_copy;
        if (bcval) {
            const size_t N = orc_npts(bcval) * (size_t)bcval->nc;
            if (!bcv_copy.p || orc_npts(&bcv_copy) * (size_t)bcv_copy.nc != N) { if (bcv_copy.p) free(bcv_copy.p); bcv_copy = *bcval; bcv_copy.p = (double*)malloc(N * sizeof(double)); }
            for (int d = 0; d < 3; ++d) { bcv_copy.lo[d] = bcval->lo[d]; bcv_copy.hi[d] = bcval->hi[d]; }
            bcv_copy.nc = bcval->nc;
            memcpy(bcv_copy.p, bcval->p, N * sizeof(double));
            g_bc_bcval = &bcv_copy;
        } else g_bc_bcval = NULL;
    }
    orc_fill_periodic(phi, g, ORC_CELL);
    for (int d = 0; d < 3; ++d) {
        if (g->periodic[d]) continue;
        for (int side = 0; side < 2; ++side) {
            const int s = 1 - 2 * side;
            const int ig = side == 0 ? -1 : g->n[d];
            const int blen = g->n[d];
            int NX = blen + 1 < maxorder ? blen + 1 : maxorder;
            double x[4] = {0.0, 0.5, 1.5, 2.5}, c[4] = {0, 0, 0, 0};
            if (NX >= 2) poly_interp_coeff(-0.5, x, NX, c);
            int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
            for (int n = 0; n < L->ncomp; ++n) {
            const int bct = side == 0 ? lobc[BCOFF(L, n) + d] : hibc[BCOFF(L, n) + d];
            for (int q2 = 0; q2 < g->n[d2]; ++q2)
            for (int q1 = 0; q1 < g->n[d1]; ++q1) {
                int idx[3]; idx[d] = ig; idx[d1] = q1; idx[d2] = q2;
                double v;
                if (bct == ORC_LO_NEUMANN || bct == ORC_LO_REFLECT_ODD) {
                    int s1[3] = {idx[0], idx[1], idx[2]}; s1[d] = ig + s;
                    v = A4(phi, s1[0], s1[1], s1[2], n);
                    if (bct == ORC_LO_REFLECT_ODD) v = -v;
                } else if (bct == ORC_LO_DIRICHLET) {
                    double bv = (inhomog && bcval) ? A4(bcval, idx[0], idx[1], idx[2], n) : 0.0;
                    if (NX < 2) v = bv;
                    else {
                        double tmp = 0.0;
                        for (int m = 1; m < NX; ++m) {
                            int sm[3] = {idx[0], idx[1], idx[2]}; sm[d] = ig + m * s;
                            tmp += A4(phi, sm[0], sm[1], sm[2], n) * c[m];
                        }
                        v = tmp + bv * c[0];
                    }
                } else continue;
                A4(phi, idx[0], idx[1], idx[2], n) = v;
            }
            }
        }
    }
    if (orc_abec_is_tensor(L)) orc_tensor_fill_edges_corners(L, phi, lobc, hibc, maxorder, inhomog, bcval);
}

void orc_abec_gsrb(const orc_abec_level* L, orc_fab* phi, const orc_fab* rhs, int redblack, double omega,
                   const int lobc[3], const int hibc[3], int maxorder)
{
    if (L->nbox > 0) { cf_gsrb(L, phi, rhs, redblack, omega, lobc, hibc, maxorder); return; }
    const orc_geom* g = &L->g;
    const double dhx = L->beta / (g->dx[0] * g->dx[0]);
    const double dhy = L->beta / (g->dx[1] * g->dx[1]);
    const double dhz = L->beta / (g->dx[2] * g->dx[2]);
    const orc_fab *bX = &L->b[0], *bY = &L->b[1], *bZ = &L->b[2];
    for (int n = 0; n < L->ncomp; ++n) {
    double cflo[3], cfhi[3];
    for (int d = 0; d < 3; ++d) {
        cflo[d] = g->periodic[d] ? 0.0 : bc_coef0(lobc[BCOFF(L, n) + d], g->n[d], maxorder);
        cfhi[d] = g->periodic[d] ? 0.0 : bc_coef0(hibc[BCOFF(L, n) + d], g->n[d], maxorder);
    }
    /* cells of one colour are independent: same result for any thread count */
    _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
    for (int k = 0; k < g->n[2]; ++k)
    for (int j = 0; j < g->n[1]; ++j)
    for (int i = 0; i < g->n[0]; ++i) {
        if ((i + j + k + redblack) % 2 != 0) continue;
        double cf0 = (i == 0) ? cflo[0] : 0.0, cf3 = (i == g->n[0] - 1) ? cfhi[0] : 0.0;
        double cf1 = (j == 0) ? cflo[1] : 0.0, cf4 = (j == g->n[1] - 1) ? cfhi[1] : 0.0;
        double cf2 = (k == 0) ? cflo[2] : 0.0, cf5 = (k == g->n[2] - 1) ? cfhi[2] : 0.0;
        double aa = (L->alpha != 0.0 && L->a.p) ? L->alpha * A4(&L->a, i, j, k, 0) : 0.0;
        double gamma = aa + dhx * (A4(bX, i, j, k, n) + A4(bX, i + 1, j, k, n))
                          + dhy * (A4(bY, i, j, k, n) + A4(bY, i, j + 1, k, n))
                          + dhz * (A4(bZ, i, j, k, n) + A4(bZ, i, j, k + 1, n));
        double g_m_d = gamma - (dhx * (A4(bX, i, j, k, n) * cf0 + A4(bX, i + 1, j, k, n) * cf3)
                              + dhy * (A4(bY, i, j, k, n) * cf1 + A4(bY, i, j + 1, k, n) * cf4)
                              + dhz * (A4(bZ, i, j, k, n) * cf2 + A4(bZ, i, j, k + 1, n) * cf5));
        double rho = dhx * (A4(bX, i, j, k, n) * A4(phi, i - 1, j, k, n) + A4(bX, i + 1, j, k, n) * A4(phi, i + 1, j, k, n))
                   + dhy * (A4(bY, i, j, k, n) * A4(phi, i, j - 1, k, n) + A4(bY, i, j + 1, k, n) * A4(phi, i, j + 1, k, n))
                   + dhz * (A4(bZ, i, j, k, n) * A4(phi, i, j, k - 1, n) + A4(bZ, i, j, k + 1, n) * A4(phi, i, j, k + 1, n));
        double res = A4(rhs, i, j, k, n) - (gamma * A4(phi, i, j, k, n) - rho);
        A4(phi, i, j, k, n) = A4(phi, i, j, k, n) + omega / g_m_d * res;
    }
    }
}

/* ---------------------------------------------------------------- transfers ---- */
void orc_cc_restrict(orc_fab* crse, const orc_fab* fine, const int cn[3])
{
    for (int n = 0; n < crse->nc; ++n)
    for (int k = 0; k < cn[2]; ++k)
    for (int j = 0; j < cn[1]; ++j)
    for (int i = 0; i < cn[0]; ++i) {
        double c = 0.0;
        for (int kr = 0; kr < 2; ++kr)
        for (int jr = 0; jr < 2; ++jr)
        for (int ir = 0; ir < 2; ++ir)
            c += A4(fine, 2 * i + ir, 2 * j + jr, 2 * k + kr, n);
        A4(crse, i, j, k, n) = 0.125 * c;
    }
}

void orc_cc_prolong_add(orc_fab* fine, const orc_fab* crse, const int fn[3])
{
    for (int n = 0; n < fine->nc; ++n)
    for (int k = 0; k < fn[2]; ++k)
    for (int j = 0; j < fn[1]; ++j)
    for (int i = 0; i < fn[0]; ++i)
        A4(fine, i, j, k, n) += A4(crse, i >> 1, j >> 1, k >> 1, n);
}

void orc_face_avgdown(orc_fab* crse, const orc_fab* fine, int dir, const int cn[3])
{
    int hi[3] = {cn[0] - 1, cn[1] - 1, cn[2] - 1};
    hi[dir] += 1;
    int d1 = (dir + 1) % 3, d2 = (dir + 2) % 3;
    for (int n = 0; n < crse->nc; ++n)
    for (int k = 0; k <= hi[2]; ++k)
    for (int j = 0; j <= hi[1]; ++j)
    for (int i = 0; i <= hi[0]; ++i) {
        int c[3] = {i, j, k};
        double s = 0.0;
        /* loop order: slower transverse index outer (kref, then jref in amrex_avgdown_faces) */
        int da = d1 < d2 ? d1 : d2, db = d1 < d2 ? d2 : d1;
        for (int rb = 0; rb < 2; ++rb)
        for (int ra = 0; ra < 2; ++ra) {
            int f[3];
            f[dir] = 2 * c[dir]; f[da] = 2 * c[da] + ra; f[db] = 2 * c[db] + rb;
            s += A4(fine, f[0], f[1], f[2], n);
        }
        A4(crse, i, j, k, n) = s * 0.25;
    }
}


/* InterpBndryData::setBndryValues, max_order 3 (interpbndrydata_{x,y,z}_o3) */
void orc_cf_interp_bndry(const orc_abec_level* L, int ratio, const orc_fab* cphi, orc_fab* bcval)
{
    orc_setval(bcval, 0.0);
    const int r = ratio;
    for (int b = 0; b < L->nbox; ++b) {
        const int* bx = L->boxes + 6 * b;
        for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
            const int t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;       /* tangential directions, ascending */
            int lo[3] = {bx[0], bx[1], bx[2]}, hi[3] = {bx[3], bx[4], bx[5]};
            lo[d] = hi[d] = side == 0 ? bx[d] - 1 : bx[3 + d] + 1;
            for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) {
                if (box_of(L, i, j, k) != 0) continue;                 /* covered or outside the domain: not a coarse/fine ghost cell */
                const int q[3] = {i, j, k};
                int c[3];
                for (int e = 0; e < 3; ++e) c[e] = q[e] >= 0 ? q[e] / r : -((-q[e] + r - 1) / r);
                /* not_covered mask of the neighbouring ghost cells (same ghost layer) */
                int m1m, m1p, m2m, m2p, mmm, mpm, mmp, mpp;
                { int a[3] = {i, j, k}; a[t1] -= r; m1m = box_of(L, a[0], a[1], a[2]) == 0; a[t1] += 2 * r; m1p = box_of(L, a[0], a[1], a[2]) == 0; }
                { int a[3] = {i, j, k}; a[t2] -= r; m2m = box_of(L, a[0], a[1], a[2]) == 0; a[t2] += 2 * r; m2p = box_of(L, a[0], a[1], a[2]) == 0; }
                { int a[3] = {i, j, k}; a[t1] -= r; a[t2] -= r; mmm = box_of(L, a[0], a[1], a[2]) == 0; a[t1] += 2 * r; mpm = box_of(L, a[0], a[1], a[2]) == 0;
                  a[t2] += 2 * r; mpp = box_of(L, a[0], a[1], a[2]) == 0; a[t1] -= 2 * r; mmp = box_of(L, a[0], a[1], a[2]) == 0; }
                for (int n = 0; n < L->ncomp; ++n) {
#define CC(o1, o2) ({ int z[3] = {c[0], c[1], c[2]}; z[t1] += (o1); z[t2] += (o2); A4(cphi, z[0], z[1], z[2], n); })
                    const int l1 = m1m ? -1 : 0, h1 = m1p ? 1 : 0, l2 = m2m ? -1 : 0, h2 = m2p ? 1 : 0;
                    const double f1 = (h1 == l1 + 1) ? 1.0 : 0.5, f2 = (h2 == l2 + 1) ? 1.0 : 0.5;
                    const double d1 = f1 * (CC(h1, 0) - CC(l1, 0));
                    const double d11 = (h1 == l1 + 2) ? 0.5 * (CC(1, 0) - 2.0 * CC(0, 0) + CC(-1, 0)) : 0.0;
                    const double d2 = f2 * (CC(0, h2) - CC(0, l2));
                    const double d22 = (h2 == l2 + 2) ? 0.5 * (CC(0, 1) - 2.0 * CC(0, 0) + CC(0, -1)) : 0.0;
                    const double d12 = (mmm && mpm && mmp && mpp) ? 0.25 * (CC(1, 1) - CC(-1, 1) + CC(-1, -1) - CC(1, -1)) : 0.0;
                    const double y1 = -0.5 + (q[t1] - c[t1] * r + 0.5) / r, y2 = -0.5 + (q[t2] - c[t2] * r + 0.5) / r;
                    A4(bcval, i, j, k, n * 3 + d) = CC(0, 0) + y1 * d1 + (y1 * y1) * d11 + y2 * d2 + (y2 * y2) * d22 + y1 * y2 * d12;
#undef CC
                }
            }
        }
    }
}

/* ---------------------------------------------------------------- multigrid ---- */
typedef struct mglev {
    orc_abec_level L;
    orc_fab cor, res, rescor;
    int owns_coef;
} mglev;

static double norminf_valid(const orc_fab* f, const int n[3], int nc)
{
    double m = 0.0;
    for (int c = 0; c < nc; ++c)
    for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i) {
        double v = fabs(A4(f, i, j, k, c));
        if (v > m) m = v;
    }
    return m;
}
static double dot_valid(const orc_fab* x, const orc_fab* y, const int n[3], int nc)
{
    double s = 0.0;
    for (int c = 0; c < nc; ++c)
    for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
        s += A4(x, i, j, k, c) * A4(y, i, j, k, c);
    return s;
}
static void subtract_mean(orc_fab* f, const int n[3], int nc)
{
    for (int c = 0; c < nc; ++c) {
        double s = 0.0;
        for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i) s += A4(f, i, j, k, c);
        double off = s / ((double)n[0] * n[1] * n[2]);
        for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i) A4(f, i, j, k, c) -= off;
    }
}
/* dst(valid) = a(valid) + s*b(valid) */
static void sxay(orc_fab* dst, const orc_fab* a, double s, const orc_fab* b, const int n[3], int nc)
{
    for (int c = 0; c < nc; ++c)
    for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
        A4(dst, i, j, k, c) = A4(a, i, j, k, c) + s * A4(b, i, j, k, c);
}
static void copy_valid(orc_fab* dst, const orc_fab* src, const int n[3], int nc)
{
    for (int c = 0; c < nc; ++c)
    for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
        A4(dst, i, j, k, c) = A4(src, i, j, k, c);
}

static int is_singular(const orc_abec_level* L, const int lobc[3], const int hibc[3])
{
    if (L->nbox > 0) return 0;                 /* coarse/fine faces carry Dirichlet data */
    if (L->alpha != 0.0 && L->a.p) return 0;
    for (int d = 0; d < 3; ++d) {
        if (L->g.periodic[d]) continue;
        for (int n = 0; n < (L->bc_percomp ? L->ncomp : 1); ++n)
            if (lobc[3 * n + d] == ORC_LO_DIRICHLET || hibc[3 * n + d] == ORC_LO_DIRICHLET) return 0;
    }
    return 1;
}

static void smooth(const mglev* m, orc_fab* sol, const orc_fab* rhs, const int lobc[3], const int hibc[3],
                   const orc_mg_opts* o, int skip_fill)
{
    g_cf_inhomog = 0; g_cf_maxorder = o->maxorder;          /* corrections: homogeneous coarse/fine data (also when the fill is skipped) */
    for (int rb = 0; rb < 2; ++rb) {
        if (!skip_fill) orc_abec_applybc(&m->L, sol, lobc, hibc, o->maxorder, 0, NULL);
        orc_abec_gsrb(&m->L, sol, rhs, rb, o->omega, lobc, hibc, o->maxorder);
        skip_fill = 0;
    }
}

/* r = b - L(x), homogeneous BC */
static void corr_residual(const mglev* m, orc_fab* r, orc_fab* x, const orc_fab* b,
                          const int lobc[3], const int hibc[3], const orc_mg_opts* o)
{
    orc_abec_applybc(&m->L, x, lobc, hibc, o->maxorder, 0, NULL);
    orc_abec_apply(&m->L, r, x);
    const int* n = m->L.g.n;
    for (int c = 0; c < m->L.ncomp; ++c)
    for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
        A4(r, i, j, k, c) = A4(b, i, j, k, c) - A4(r, i, j, k, c);
    if (m->L.nbox > 0) cf_zero_uncovered(&m->L, r);
}

/* MLCGSolver::solve_bicgstab */
static int bicgstab(const mglev* m, orc_fab* sol, const orc_fab* rhs, const int lobc[3], const int hibc[3],
                    const orc_mg_opts* o, double eps_rel, double eps_abs, int* niters)
{
    const int* n = m->L.g.n;
    const int nc = m->L.ncomp;
    orc_fab ph = orc_alloc(n, ORC_CELL, 1, nc), sh = orc_alloc(n, ORC_CELL, 1, nc);
    orc_fab sorig = orc_alloc(n, ORC_CELL, 0, nc), p = orc_alloc(n, ORC_CELL, 0, nc), r = orc_alloc(n, ORC_CELL, 0, nc);
    orc_fab s = orc_alloc(n, ORC_CELL, 0, nc), rh = orc_alloc(n, ORC_CELL, 0, nc), v = orc_alloc(n, ORC_CELL, 0, nc), t = orc_alloc(n, ORC_CELL, 0, nc);
    corr_residual(m, &r, sol, rhs, lobc, hibc, o);
    copy_valid(&sorig, sol, n, nc);
    copy_valid(&rh, &r, n, nc);
    orc_setval(sol, 0.0);
    double rnorm = norminf_valid(&r, n, nc);
    const double rnorm0 = rnorm;
    int ret = 0, nit = 1;
    double rho_1 = 0, alpha = 0, omega = 0;
    if (rnorm0 == 0 || rnorm0 < eps_abs) { nit = 0; goto done; }
    /* Krylov bound: at most 2N iterations for N unknowns (mirrors the product; beyond N only round-off is chased) */
    long nunk = (long)n[0] * n[1] * n[2] * nc;
    long cap = 2 * nunk < 8 ? 8 : 2 * nunk;
    const int maxiter = (int)(o->bottom_maxiter < cap ? o->bottom_maxiter : cap);
    for (; nit <= maxiter; ++nit) {
        const double rho = dot_valid(&rh, &r, n, nc);
        if (rho == 0) { ret = 1; break; }
        if (nit == 1) copy_valid(&p, &r, n, nc);
        else {
            const double beta = (rho / rho_1) * (alpha / omega);
            sxay(&p, &p, -omega, &v, n, nc);
            sxay(&p, &r, beta, &p, n, nc);
        }
        copy_valid(&ph, &p, n, nc);
        orc_abec_applybc(&m->L, &ph, lobc, hibc, o->maxorder, 0, NULL);
        orc_abec_apply(&m->L, &v, &ph);
        const double rhTv = dot_valid(&rh, &v, n, nc);
        if (rhTv != 0) alpha = rho / rhTv; else { ret = 2; break; }
        sxay(sol, sol, alpha, &ph, n, nc);
        sxay(&s, &r, -alpha, &v, n, nc);
        rnorm = norminf_valid(&s, n, nc);
        if (rnorm < eps_rel * rnorm0 || rnorm < eps_abs) break;
        copy_valid(&sh, &s, n, nc);
        orc_abec_applybc(&m->L, &sh, lobc, hibc, o->maxorder, 0, NULL);
        orc_abec_apply(&m->L, &t, &sh);
        const double tt = dot_valid(&t, &t, n, nc), ts = dot_valid(&t, &s, n, nc);
        if (tt != 0) omega = ts / tt; else { ret = 3; break; }
        sxay(sol, sol, omega, &sh, n, nc);
        sxay(&r, &s, -omega, &t, n, nc);
        rnorm = norminf_valid(&r, n, nc);
        if (rnorm < eps_rel * rnorm0 || rnorm < eps_abs) break;
        if (omega == 0) { ret = 4; break; }
        rho_1 = rho;
    }
    if (ret == 0 && rnorm > eps_rel * rnorm0 && rnorm > eps_abs) ret = 8;
    if ((ret == 0 || ret == 8) && rnorm < rnorm0) sxay(sol, sol, 1.0, &sorig, n, nc);
    else { orc_setval(sol, 0.0); sxay(sol, sol, 1.0, &sorig, n, nc); }
done:
    if (niters) *niters = nit;
    orc_free(&ph); orc_free(&sh); orc_free(&sorig); orc_free(&p); orc_free(&r); orc_free(&s); orc_free(&rh); orc_free(&v); orc_free(&t);
    return ret;
}

static void bottom_solve(mglev* m, const int lobc[3], const int hibc[3], const orc_mg_opts* o, int singular, orc_mg_stats* st)
{
    const int* n = m->L.g.n;
    orc_setval(&m->cor, 0.0);
    if (o->bottom_smoother_only) {
        int skip = 1;
        for (int i = 0; i < o->nuf; ++i) { smooth(m, &m->cor, &m->res, lobc, hibc, o, skip); skip = 0; }
        return;
    }
    orc_fab b = orc_alloc(n, ORC_CELL, 0, m->L.ncomp);
    copy_valid(&b, &m->res, n, m->L.ncomp);
    if (singular) subtract_mean(&b, n, m->L.ncomp);
    int nit = 0;
    int ret = bicgstab(m, &m->cor, &b, lobc, hibc, o, o->bottom_reltol, -1.0, &nit);
    if (st) st->bottom_iters_total += nit;
    if (ret != 0) {
        orc_setval(&m->cor, 0.0);
        int skip = 1;
        for (int i = 0; i < o->nuf; ++i) { smooth(m, &m->cor, &m->res, lobc, hibc, o, skip); skip = 0; }
    }
    const int nn = (ret == 0) ? o->nub : o->nuf;
    for (int i = 0; i < nn; ++i) smooth(m, &m->cor, &m->res, lobc, hibc, o, 0);
    orc_free(&b);
}

static void vcycle(mglev* mg, int nlev, const int lobc[3], const int hibc[3], const orc_mg_opts* o, int singular, orc_mg_stats* st)
{
    for (int l = 0; l < nlev - 1; ++l) {
        orc_setval(&mg[l].cor, 0.0);
        int skip = 1;
        for (int i = 0; i < o->nu1; ++i) { smooth(&mg[l], &mg[l].cor, &mg[l].res, lobc, hibc, o, skip); skip = 0; }
        corr_residual(&mg[l], &mg[l].rescor, &mg[l].cor, &mg[l].res, lobc, hibc, o);
        orc_cc_restrict(&mg[l + 1].res, &mg[l].rescor, mg[l + 1].L.g.n);
    }
    if (nlev == 1 && 0) {}
    bottom_solve(&mg[nlev - 1], lobc, hibc, o, singular, st);
    for (int l = nlev - 2; l >= 0; --l) {
        orc_cc_prolong_add(&mg[l].cor, &mg[l + 1].cor, mg[l].L.g.n);
        for (int i = 0; i < o->nu2; ++i) smooth(&mg[l], &mg[l].cor, &mg[l].res, lobc, hibc, o, 0);
    }
}

static int build_hierarchy(const orc_abec_level* L, mglev* mg, const orc_mg_opts* o)
{
    int nlev = 1;
    mg[0].L = *L; mg[0].owns_coef = 0;
    while (nlev <= o->max_coarsening_level && nlev < 32) {
        const orc_geom* fg = &mg[nlev - 1].L.g;
        int ok = 1;
        for (int d = 0; d < 3; ++d) if (fg->n[d] % 2 != 0 || fg->n[d] / 2 < o->min_width) ok = 0;
        if (!ok) break;
        mglev* c = &mg[nlev];
        c->L = mg[nlev - 1].L;
        if (L->nbox > 0) {
            /* the level's boxes coarsen with the multigrid; stop when one of them cannot (MLLinOp: box array not coarsenable) */
            const int* fb = mg[nlev - 1].L.boxes;
            for (int q = 0; q < L->nbox && ok; ++q)
                for (int d = 0; d < 3; ++d) {
                    const int lo = fb[6 * q + d], len = fb[6 * q + 3 + d] - lo + 1;
                    if (lo % 2 != 0 || len % 2 != 0 || len / 2 < o->min_width) ok = 0;
                }
            if (!ok) break;
            int* cb = (int*)malloc(sizeof(int) * 6 * L->nbox);
            for (int q = 0; q < L->nbox; ++q)
                for (int d = 0; d < 3; ++d) { cb[6 * q + d] = fb[6 * q + d] / 2; cb[6 * q + 3 + d] = (fb[6 * q + 3 + d] + 1) / 2 - 1; }
            c->L.boxes = cb;
        }
        for (int d = 0; d < 3; ++d) { c->L.g.n[d] = fg->n[d] / 2; c->L.g.dx[d] = fg->dx[d] * 2.0; }
        c->owns_coef = 1;
        if (L->a.p) {
            c->L.a = orc_alloc(c->L.g.n, ORC_CELL, 0, 1);
            orc_cc_restrict(&c->L.a, &mg[nlev - 1].L.a, c->L.g.n);
        }
        for (int d = 0; d < 3; ++d) {
            c->L.b[d] = orc_alloc(c->L.g.n, ORC_FACE[d], 0, L->b[d].nc);
            orc_face_avgdown(&c->L.b[d], &mg[nlev - 1].L.b[d], d, c->L.g.n);
        }
        ++nlev;
    }
    for (int l = 0; l < nlev; ++l) {
        mg[l].cor = orc_alloc(mg[l].L.g.n, ORC_CELL, 1, L->ncomp);
        mg[l].res = orc_alloc(mg[l].L.g.n, ORC_CELL, 0, L->ncomp);
        mg[l].rescor = orc_alloc(mg[l].L.g.n, ORC_CELL, 0, L->ncomp);
    }
    return nlev;
}

static void free_hierarchy(mglev* mg, int nlev)
{
    for (int l = 0; l < nlev; ++l) {
        orc_free(&mg[l].cor); orc_free(&mg[l].res); orc_free(&mg[l].rescor);
        if (mg[l].owns_coef) {
            if (mg[l].L.nbox > 0) free((void*)mg[l].L.boxes);
            if (mg[l].L.a.p) orc_free(&mg[l].L.a);
            for (int d = 0; d < 3; ++d) orc_free(&mg[l].L.b[d]);
        }
    }
}

void orc_abec_solve_cf(const orc_abec_level* L, orc_fab* phi, const orc_fab* rhs, const int lobc[3], const int hibc[3],
                       const orc_fab* cf_bcval, double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st)
{
    g_cf_bcval = cf_bcval;
    orc_abec_solve(L, phi, rhs, lobc, hibc, rtol, atol, o, st);
    g_cf_bcval = NULL;
}

void orc_abec_solve(const orc_abec_level* L, orc_fab* phi, const orc_fab* rhs_in,
                    const int lobc[3], const int hibc[3], double rtol, double atol,
                    const orc_mg_opts* o, orc_mg_stats* st)
{
    mglev mg[32];
    memset(mg, 0, sizeof(mg));
    const int nlev = build_hierarchy(L, mg, o);
    const int* n = L->g.n;
    const int nc = L->ncomp;
    const int singular = is_singular(L, lobc, hibc);
    orc_mg_stats loc; memset(&loc, 0, sizeof(loc));

    orc_fab rhs = orc_alloc(n, ORC_CELL, 0, nc);
    copy_valid(&rhs, rhs_in, n, nc);
    if (L->nbox > 0) cf_zero_uncovered(L, &rhs);
    if (singular) subtract_mean(&rhs, n, nc);

    /* inhomogeneous BC data = ghost values of phi on entry (MLMG setLevelBC) */
    orc_fab bcval = orc_alloc(n, ORC_CELL, 1, nc);
    orc_copy_all(&bcval, phi);

    orc_fab* res = &mg[0].res;
    orc_abec_applybc(L, phi, lobc, hibc, o->maxorder, 1, &bcval);
    orc_abec_apply(L, res, phi);
    for (int c = 0; c < nc; ++c)
    for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
        A4(res, i, j, k, c) = A4(&rhs, i, j, k, c) - A4(res, i, j, k, c);
    if (L->nbox > 0) cf_zero_uncovered(L, res);

    loc.resnorm0 = norminf_valid(res, n, nc);
    loc.rhsnorm0 = norminf_valid(&rhs, n, nc);
    const double max_norm = loc.rhsnorm0 >= loc.resnorm0 ? loc.rhsnorm0 : loc.resnorm0;
    const double res_target = fmax(atol, fmax(rtol, 1.e-16) * max_norm);
    loc.resnorm = loc.resnorm0;
    if (o->verbose) printf("orc MLMG: initial rhs %.6e resid0 %.6e target %.3e levels %d\n", loc.rhsnorm0, loc.resnorm0, res_target, nlev);
    if (o->fixed_iters <= 0 && loc.resnorm0 <= res_target) loc.converged = 1;
    else {
        const int maxit = o->fixed_iters > 0 ? o->fixed_iters : o->max_iters;
        for (int iter = 0; iter < maxit; ++iter) {
            if (singular) subtract_mean(res, n, nc);
            vcycle(mg, nlev, lobc, hibc, o, singular, &loc);
            for (int c = 0; c < nc; ++c)
            for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
                A4(phi, i, j, k, c) += A4(&mg[0].cor, i, j, k, c);
            orc_abec_applybc(L, phi, lobc, hibc, o->maxorder, 1, &bcval);
            orc_abec_apply(L, res, phi);
            for (int c = 0; c < nc; ++c)
            for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) for (int i = 0; i < n[0]; ++i)
                A4(res, i, j, k, c) = A4(&rhs, i, j, k, c) - A4(res, i, j, k, c);
            if (L->nbox > 0) cf_zero_uncovered(L, res);
            loc.resnorm = norminf_valid(res, n, nc);
            loc.iters = iter + 1;
            if (o->verbose) printf("orc MLMG: iter %d resid %.6e ratio %.3e\n", iter + 1, loc.resnorm, loc.resnorm / max_norm);
            if (o->fixed_iters <= 0 && loc.resnorm <= res_target) { loc.converged = 1; break; }
        }
    }
    /* final ghost fill of the solution (setFinalFillBC-like; harmless for callers that refill) */
    orc_abec_applybc(L, phi, lobc, hibc, o->maxorder, 1, &bcval);
    if (st) *st = loc;
    orc_free(&rhs); orc_free(&bcval);
    free_hierarchy(mg, nlev);
}

void orc_abec_flux(const orc_abec_level* L, orc_fab* flux[3], const orc_fab* phi)
{
    const orc_geom* g = &L->g;
    for (int d = 0; d < 3; ++d) {
        const double fac = L->beta / g->dx[d];
        int hi[3] = {g->n[0] - 1, g->n[1] - 1, g->n[2] - 1};
        hi[d] += 1;
        for (int n = 0; n < L->ncomp; ++n)
        for (int k = 0; k <= hi[2]; ++k) for (int j = 0; j <= hi[1]; ++j) for (int i = 0; i <= hi[0]; ++i) {
            int m[3] = {i, j, k}; m[d] -= 1;
            A4(flux[d], i, j, k, n) = -fac * A4(&L->b[d], i, j, k, n) * (A4(phi, i, j, k, n) - A4(phi, m[0], m[1], m[2], n));
        }
    }
}

/* Diffusion::computeExtensiveFluxes on the ABec part of an operator (Source/Diffusion.cpp:1463-1537: MLMG::getFluxes = the face fluxes
 * WITHOUT the b scalar, times fac x face area): flux_d(n) = or += fac * area_d * ( -b_d(n) dphi_n/dx_d ) on every face of the level.
 * phi: ghost cells outside the physical domain as the operator's applyBC left them; on a partial level (nbox > 0) the coarse/fine
 * state of the solve (orc_cf_set_bcval) must be set: coarse/fine faces use the ghost formula. */
void orc_abec_extensive_flux(const orc_abec_level* L, orc_fab* flux[3], const orc_fab* phi, double fac, int add)
{
    const orc_geom* g = &L->g;
    for (int d = 0; d < 3; ++d) {
        const double sc = fac * g->dx[(d + 1) % 3] * g->dx[(d + 2) % 3] / g->dx[d];
        orc_fab t = orc_alloc(g->n, ORC_FACE[d], 0, L->ncomp);
        orc_setval(&t, 0.0);
        if (L->nbox == 0) {
            int hi[3] = {g->n[0] - 1, g->n[1] - 1, g->n[2] - 1};
            hi[d] += 1;
            for (int n = 0; n < L->ncomp; ++n)
            for (int k = 0; k <= hi[2]; ++k) for (int j = 0; j <= hi[1]; ++j) for (int i = 0; i <= hi[0]; ++i) {
                int m[3] = {i, j, k}; m[d] -= 1;
                A4(&t, i, j, k, n) = -A4(&L->b[d], i, j, k, n) * (A4(phi, i, j, k, n) - A4(phi, m[0], m[1], m[2], n));
            }
        } else {
            for (int b = 0; b < L->nbox; ++b) {
                const int* bx = L->boxes + 6 * b;
                for (int n = 0; n < L->ncomp; ++n)
                for (int k = bx[2]; k <= bx[5]; ++k) for (int j = bx[1]; j <= bx[4]; ++j) for (int i = bx[0]; i <= bx[3]; ++i) {
                    const int idx[3] = {i, j, k};
                    int cf;
                    {   /* low face of the cell */
                        const double xm = box_nbr(L, phi, bx, i, j, k, n, d, -1, &cf);
                        A4(&t, i, j, k, n) = -A4(&L->b[d], i, j, k, n) * (A4(phi, i, j, k, n) - xm);
                    }
                    if (idx[d] == bx[3 + d]) {          /* high face of the last cell */
                        int f[3] = {i, j, k}; f[d] += 1;
                        const double xp = box_nbr(L, phi, bx, i, j, k, n, d, +1, &cf);
                        A4(&t, f[0], f[1], f[2], n) = -A4(&L->b[d], f[0], f[1], f[2], n) * (xp - A4(phi, i, j, k, n));
                    }
                }
            }
        }
        const size_t N = orc_npts(&t) * (size_t)L->ncomp;
        for (size_t q = 0; q < N; ++q) flux[d]->p[q] = (add ? flux[d]->p[q] : 0.0) + sc * t.p[q];
        orc_free(&t);
    }
}

/* ---------------------------------------------------------------- MAC projection */
void orc_mac_divergence(const orc_geom* g, orc_fab* div, orc_fab* const umac[3])
{
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(div, i, j, k, 0) = (A4(umac[0], i + 1, j, k, 0) - A4(umac[0], i, j, k, 0)) / g->dx[0]
                            + (A4(umac[1], i, j + 1, k, 0) - A4(umac[1], i, j, k, 0)) / g->dx[1]
                            + (A4(umac[2], i, j, k + 1, 0) - A4(umac[2], i, j, k, 0)) / g->dx[2];
}

void orc_mac_project(const orc_geom* g, orc_fab* umac[3], const orc_fab* rho, const orc_fab* S, orc_fab* phi,
                     double rhs_scale, const int lobc[3], const int hibc[3], double rtol, double atol,
                     const orc_mg_opts* o, orc_mg_stats* st)
{
    orc_abec_level L;
    memset(&L, 0, sizeof(L));
    L.g = *g; L.alpha = 0.0; L.beta = 1.0; L.ncomp = 1; L.a.p = NULL;
    /* average_cellcenter_to_face(rho) then invert(1/rhs_scale): b = (1/rhs_scale)/rho_face
     * (reference Source/MacProj.cpp:1115-1127) */
    const double scale = 1.0 / rhs_scale;
    for (int d = 0; d < 3; ++d) {
        L.b[d] = orc_alloc(g->n, ORC_FACE[d], 0, 1);
        int hi[3] = {g->n[0] - 1, g->n[1] - 1, g->n[2] - 1}; hi[d] += 1;
        for (int k = 0; k <= hi[2]; ++k) for (int j = 0; j <= hi[1]; ++j) for (int i = 0; i <= hi[0]; ++i) {
            int m[3] = {i, j, k}; m[d] -= 1;
            double rf = 0.5 * (A4(rho, m[0], m[1], m[2], 0) + A4(rho, i, j, k, 0));
            A4(&L.b[d], i, j, k, 0) = scale / rf;
        }
    }
    orc_fab rhs = orc_alloc(g->n, ORC_CELL, 0, 1);
    orc_mac_divergence(g, &rhs, umac);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        double v = -A4(&rhs, i, j, k, 0);
        if (S) v += A4(S, i, j, k, 0);
        A4(&rhs, i, j, k, 0) = v;
    }
    orc_abec_solve(&L, phi, &rhs, lobc, hibc, rtol, atol, o, st);
    orc_fab fl[3]; orc_fab* flp[3];
    for (int d = 0; d < 3; ++d) { fl[d] = orc_alloc(g->n, ORC_FACE[d], 0, 1); flp[d] = &fl[d]; }
    orc_abec_flux(&L, flp, phi);
    for (int d = 0; d < 3; ++d) {
        int hi[3] = {g->n[0] - 1, g->n[1] - 1, g->n[2] - 1}; hi[d] += 1;
        for (int k = 0; k <= hi[2]; ++k) for (int j = 0; j <= hi[1]; ++j) for (int i = 0; i <= hi[0]; ++i)
            A4(umac[d], i, j, k, 0) += A4(&fl[d], i, j, k, 0);
        orc_free(&fl[d]); orc_free(&L.b[d]);
    }
    orc_free(&rhs);
}

void orc_mac_project_cf(const orc_geom* g, orc_fab* umac[3], const orc_fab* rho, const orc_fab* S, orc_fab* phi, double rhs_scale,
                        const int lobc[3], const int hibc[3], int nbox, const int* boxes, int ratio, const orc_fab* cphi,
                        double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st)
{
    orc_abec_level L;
    memset(&L, 0, sizeof(L));
    L.g = *g; L.alpha = 0.0; L.beta = 1.0; L.ncomp = 1; L.a.p = NULL;
    L.nbox = nbox; L.boxes = boxes;
    for (int d = 0; d < 3; ++d) L.cf_loc[d] = 0.5 * ratio * g->dx[d];
    const double scale = 1.0 / rhs_scale;
    orc_fab rhs = orc_alloc(g->n, ORC_CELL, 0, 1);
    orc_setval(&rhs, 0.0);
    for (int d = 0; d < 3; ++d) { L.b[d] = orc_alloc(g->n, ORC_FACE[d], 0, 1); orc_setval(&L.b[d], 0.0); }
    for (int b = 0; b < nbox; ++b) {
        const int* bx = boxes + 6 * b;
        for (int d = 0; d < 3; ++d) {
            int hi[3] = {bx[3], bx[4], bx[5]}; hi[d] += 1;
            for (int k = bx[2]; k <= hi[2]; ++k) for (int j = bx[1]; j <= hi[1]; ++j) for (int i = bx[0]; i <= hi[0]; ++i) {
                int m[3] = {i, j, k}; m[d] -= 1;
                const double rf = 0.5 * (A4(rho, m[0], m[1], m[2], 0) + A4(rho, i, j, k, 0));
                A4(&L.b[d], i, j, k, 0) = scale / rf;
            }
        }
        for (int k = bx[2]; k <= bx[5]; ++k) for (int j = bx[1]; j <= bx[4]; ++j) for (int i = bx[0]; i <= bx[3]; ++i) {
            double dv = (A4(umac[0], i + 1, j, k, 0) - A4(umac[0], i, j, k, 0)) / g->dx[0]
                      + (A4(umac[1], i, j + 1, k, 0) - A4(umac[1], i, j, k, 0)) / g->dx[1]
                      + (A4(umac[2], i, j, k + 1, 0) - A4(umac[2], i, j, k, 0)) / g->dx[2];
            A4(&rhs, i, j, k, 0) = (S ? A4(S, i, j, k, 0) : 0.0) - dv;
        }
    }
    orc_fab bcv = orc_alloc(g->n, ORC_CELL, 1, 3);
    orc_cf_interp_bndry(&L, ratio, cphi, &bcv);
    orc_abec_solve_cf(&L, phi, &rhs, lobc, hibc, &bcv, rtol, atol, o, st);
    /* u_mac -= b grad phi on every face of the level, each face once (a face shared by two boxes sees the same two cells) */
    g_cf_bcval = &bcv; g_cf_inhomog = 1; g_cf_maxorder = o->maxorder;
    for (int d = 0; d < 3; ++d) {
        orc_fab done = orc_alloc(g->n, ORC_FACE[d], 0, 1);
        orc_setval(&done, 0.0);
        const double fac = L.beta / g->dx[d];
        for (int b = 0; b < nbox; ++b) {
            const int* bx = boxes + 6 * b;
            for (int k = bx[2]; k <= bx[5]; ++k) for (int j = bx[1]; j <= bx[4]; ++j) for (int i = bx[0]; i <= bx[3]; ++i) {
                const int idx[3] = {i, j, k};
                for (int side = 0; side < 2; ++side) {
                    if (side == 1 && idx[d] != bx[3 + d]) continue;            /* the high face of the last cell only */
                    int f[3] = {i, j, k}, cf;
                    if (side == 1) f[d] += 1;
                    if (A4(&done, f[0], f[1], f[2], 0) != 0.0) continue;
                    const double xn = box_nbr(&L, phi, bx, i, j, k, 0, d, side == 0 ? -1 : 1, &cf);
                    const double dphi = side == 0 ? A4(phi, i, j, k, 0) - xn : xn - A4(phi, i, j, k, 0);
                    A4(umac[d], f[0], f[1], f[2], 0) += -fac * A4(&L.b[d], f[0], f[1], f[2], 0) * dphi;
                    A4(&done, f[0], f[1], f[2], 0) = 1.0;
                    /* the periodic image of a face on the domain boundary */
                    if (g->periodic[d] && (f[d] == 0 || f[d] == g->n[d])) {
                        int f2[3] = {f[0], f[1], f[2]}; f2[d] = f[d] == 0 ? g->n[d] : 0;
                        if (A4(&done, f2[0], f2[1], f2[2], 0) == 0.0) { A4(umac[d], f2[0], f2[1], f2[2], 0) = A4(umac[d], f[0], f[1], f[2], 0); A4(&done, f2[0], f2[1], f2[2], 0) = 1.0; }
                    }
                }
            }
        }
        orc_free(&done);
    }
    g_cf_bcval = NULL;
    orc_free(&bcv); orc_free(&rhs);
    for (int d = 0; d < 3; ++d) orc_free(&L.b[d]);
}
