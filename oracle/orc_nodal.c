/* oracle/orc_nodal.c -- nodal (Q1 finite-element) variable-sigma Laplacian, nodal multigrid and
 * the approximate (nodal) projection, restated on the CPU (test infrastructure only; PARITY
 * UNPINNED, see orc.h).
 *
 * The operator is assembled ELEMENT BY ELEMENT from the trilinear stiffness matrix (an independent
 * formulation of the 27-point stencil that AMReX writes out explicitly in mlndlap_adotx_aa), so
 * agreement with the product's explicit-stencil HIP kernels is a real check.
 *
 * Follows (upstream AMReX, not in /root/reference): MLNodeLaplacian (adotx_aa, gauss_seidel_aa /
 * gscolor_aa / jacobi_aa, restriction, interpadd_aa, divu, mknewu_aa), MLNodeLinOp, MLMG and
 * Hydro::NodalProjector::project.  Reference call sites: Source/Projection.cpp:2385-2567
 * (doMLMGNodalProjection: Gauss-Seidel on, harmonic average off, max_fmg_iter 0, proj_tol 1e-12),
 * Source/NavierStokesBase.cpp:4102-4122 (computeGradP -> compGrad).
 */
#include "orc_int.h"
void orc_nodal_fill_bc(const orc_geom* g, orc_fab* x, const int lobc[3], const int hibc[3]);
void orc_sigma_fill_bc(const orc_geom* g, orc_fab* s);

/* Neumann-like faces of the nodal operator: walls and inflow faces */
#define NEU(b) ((b) == ORC_LO_NEUMANN || (b) == ORC_LO_INFLOW)

/* local node a = (ax,ay,az) in {0,1}^3 ; weight of x_b in row a for one element with sigma=1:
 * w_ab = -[ sx/hx^2 my mz + sy/hy^2 mx mz + sz/hz^2 mx my ],  s=+1 same / -1 differ,  m=1/3 same / 1/6 differ */
static inline double elem_w(int a, int b, const double* dx)
{
    int ax = a & 1, ay = (a >> 1) & 1, az = (a >> 2) & 1;
    int bx = b & 1, by = (b >> 1) & 1, bz = (b >> 2) & 1;
    double sx = ax == bx ? 1. : -1., sy = ay == by ? 1. : -1., sz = az == bz ? 1. : -1.;
    double mx = ax == bx ? 1. / 3. : 1. / 6., my = ay == by ? 1. / 3. : 1. / 6., mz = az == bz ? 1. / 3. : 1. / 6.;
    return -(sx / (dx[0] * dx[0]) * my * mz + sy / (dx[1] * dx[1]) * mx * mz + sz / (dx[2] * dx[2]) * mx * my);
}

/* (A x)(node) and diagonal coefficient */
static inline double node_Ax(const orc_geom* g, const orc_fab* x, const orc_fab* sig, int i, int j, int k, double* diag)
{
    double y = 0.0, dg = 0.0;
    /* loop over the 8 cells touching node (i,j,k): cell (i-1+cx, j-1+cy, k-1+cz); the node is local node a=(1-cx,1-cy,1-cz) */
    for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
        int ci = i - 1 + cx, cj = j - 1 + cy, ck = k - 1 + cz;
        double s = A4(sig, ci, cj, ck, 0);
        int a = (1 - cx) | ((1 - cy) << 1) | ((1 - cz) << 2);
        for (int b = 0; b < 8; ++b) {
            int bx = b & 1, by = (b >> 1) & 1, bz = (b >> 2) & 1;
            double w = s * elem_w(a, b, g->dx);
            if (b == a) dg += w;
            else y += w * A4(x, ci + bx, cj + by, ck + bz, 0);
        }
    }
    if (diag) *diag = dg;
    return y + dg * A4(x, i, j, k, 0);
}

void orc_nodal_adotx(const orc_geom* g, orc_fab* y, const orc_fab* x, const orc_fab* sig)
{
    _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
        A4(y, i, j, k, 0) = node_Ax(g, x, sig, i, j, k, NULL);
}

/* mlndlap_divu + mlndlap_impose_neumann_bc: cells outside a Neumann wall contribute zero velocity, then the rhs of
 * wall nodes is doubled per wall direction (the operator rows there are doubled by the mirrored ghost data). */
void orc_nodal_divu_bc(const orc_geom* g, orc_fab* rhs, const orc_fab* vel, const int lobc[3], const int hibc[3])
{
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
        double r = 0.0;
        for (int d = 0; d < 3; ++d) {
            double fac = 0.25 / g->dx[d], s = 0.0;
            for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
                int c[3] = {cx, cy, cz};
                int cell[3] = {i - 1 + cx, j - 1 + cy, k - 1 + cz};
                /* set_boundary_velocity (Source/Projection.cpp:2570-2663) + mlndlap_divu: cells outside a Neumann wall carry no
                 * velocity; outside an inflow face only the normal component (the inflow value) survives */
                int outside = 0;
                for (int e = 0; e < 3; ++e) {
                    if (g->periodic[e]) continue;
                    const int bt = cell[e] < 0 ? lobc[e] : (cell[e] > g->n[e] - 1 ? hibc[e] : 0);
                    if (bt == ORC_LO_NEUMANN) outside = 1;
                    else if (bt == ORC_LO_INFLOW && e != d) outside = 1;
                }
                double sgn = c[d] ? 1.0 : -1.0;
                s += sgn * (outside ? 0.0 : A4(vel, cell[0], cell[1], cell[2], d));
            }
            r += fac * s;
        }
        const int idx[3] = {i, j, k};
        for (int e = 0; e < 3; ++e) {
            if (g->periodic[e]) continue;
            if (idx[e] == 0 && NEU(lobc[e])) r *= 2.0;
            if (idx[e] == g->n[e] && NEU(hibc[e])) r *= 2.0;
        }
        A4(rhs, i, j, k, 0) = r;
    }
}

void orc_nodal_divu(const orc_geom* g, orc_fab* rhs, const orc_fab* vel)
{
    int bc[3];
    for (int d = 0; d < 3; ++d) bc[d] = g->periodic[d] ? ORC_LO_PERIODIC : ORC_LO_NEUMANN;
    orc_nodal_divu_bc(g, rhs, vel, bc, bc);
}

void orc_nodal_mknewu(const orc_geom* g, orc_fab* vel, const orc_fab* phi, const orc_fab* sig)
{
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        for (int d = 0; d < 3; ++d) {
            double fac = 0.25 / g->dx[d], s = 0.0;
            for (int nz = 0; nz < 2; ++nz) for (int ny = 0; ny < 2; ++ny) for (int nx = 0; nx < 2; ++nx) {
                int c[3] = {nx, ny, nz};
                s += (c[d] ? 1.0 : -1.0) * A4(phi, i + nx, j + ny, k + nz, 0);
            }
            A4(vel, i, j, k, d) -= A4(sig, i, j, k, 0) * fac * s;
        }
}

void orc_nodal_compgrad(const orc_geom* g, orc_fab* gp, const orc_fab* phi)
{
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        for (int d = 0; d < 3; ++d) {
            double fac = 0.25 / g->dx[d], s = 0.0;
            for (int nz = 0; nz < 2; ++nz) for (int ny = 0; ny < 2; ++ny) for (int nx = 0; nx < 2; ++nx) {
                int c[3] = {nx, ny, nz};
                s += (c[d] ? 1.0 : -1.0) * A4(phi, i + nx, j + ny, k + nz, 0);
            }
            A4(gp, i, j, k, d) = fac * s;
        }
}

/* Ghost nodes: periodic images, then even reflection about Neumann walls (mlndlap_applybc): x(lo-m) = x(lo+m).
 * Directions are processed one after the other over the full extent of the others, so edge/corner ghosts compose. */
static void nodal_fill_bc(const orc_geom* g, orc_fab* x, const int lobc[3], const int hibc[3])
{
    orc_fill_periodic(x, g, ORC_NODE);
    for (int d = 0; d < 3; ++d) {
        if (g->periodic[d]) continue;
        for (int k = x->lo[2]; k <= x->hi[2]; ++k) for (int j = x->lo[1]; j <= x->hi[1]; ++j) for (int i = x->lo[0]; i <= x->hi[0]; ++i) {
            int idx[3] = {i, j, k}, s[3] = {i, j, k};
            if (idx[d] < 0 && lobc && NEU(lobc[d])) s[d] = -idx[d];
            else if (idx[d] > g->n[d] && hibc && NEU(hibc[d])) s[d] = 2 * g->n[d] - idx[d];
            else continue;
            A4(x, i, j, k, 0) = A4(x, s[0], s[1], s[2], 0);
        }
    }
}
void orc_nodal_fill_bc(const orc_geom* g, orc_fab* x, const int lobc[3], const int hibc[3]) { nodal_fill_bc(g, x, lobc, hibc); }
static const int PERIODIC_BC[3] = {ORC_LO_PERIODIC, ORC_LO_PERIODIC, ORC_LO_PERIODIC};
/* BCs of the solve in progress (the oracle is single-threaded) */
static const int *g_lobc = PERIODIC_BC, *g_hibc = PERIODIC_BC;
static void nodal_fill(const orc_geom* g, orc_fab* x) { nodal_fill_bc(g, x, g_lobc, g_hibc); }
/* Dirichlet node mask of the level being worked on (NULL: none): nodes with mask != 0 keep their value, carry no residual
 * and take no correction (MLNodeLaplacian dirichlet mask: outflow faces and the boundary of a level that does not cover the
 * domain) */
static const orc_fab* g_dm = NULL;
static inline int dm_on(int i, int j, int k) { return g_dm && A4(g_dm, i, j, k, 0) != 0.0; }

/* sigma ghost cells: periodic images, mirror across non-periodic walls (mlndlap_fillbc_cc): sig(lo-m) = sig(lo+m-1) */
static void sigma_fill_bc(const orc_geom* g, orc_fab* s)
{
    orc_fill_periodic(s, g, ORC_CELL);
    for (int d = 0; d < 3; ++d) {
        if (g->periodic[d]) continue;
        for (int k = s->lo[2]; k <= s->hi[2]; ++k) for (int j = s->lo[1]; j <= s->hi[1]; ++j) for (int i = s->lo[0]; i <= s->hi[0]; ++i) {
            int idx[3] = {i, j, k}, q[3] = {i, j, k};
            if (idx[d] < 0) q[d] = -idx[d] - 1;
            else if (idx[d] > g->n[d] - 1) q[d] = 2 * g->n[d] - 1 - idx[d];
            else continue;
            A4(s, i, j, k, 0) = A4(s, q[0], q[1], q[2], 0);
        }
    }
}

void orc_sigma_fill_bc(const orc_geom* g, orc_fab* s) { sigma_fill_bc(g, s); }

/* weight of a node in sums / dot products: 0 for periodic duplicates, 1/2 per Neumann wall the node lies on
 * (the nodal system is stored in the "doubled" form A_full = 2^k A_half at wall nodes; MLNodeLinOp dot mask) */
static inline double node_weight(const orc_geom* g, const int lobc[3], const int hibc[3], int i, int j, int k)
{
    const int idx[3] = {i, j, k};
    double w = 1.0;
    for (int d = 0; d < 3; ++d) {
        if (g->periodic[d]) { if (idx[d] == g->n[d]) return 0.0; }
        else {
            if (idx[d] == 0 && NEU(lobc[d])) w *= 0.5;
            if (idx[d] == g->n[d] && NEU(hibc[d])) w *= 0.5;
        }
    }
    return w;
}

void orc_nodal_smooth(const orc_geom* g, orc_fab* x, const orc_fab* rhs, const orc_fab* sig, int smoother, int nsweeps,
                      const int lobc[3], const int hibc[3])
{
    g_lobc = lobc ? lobc : PERIODIC_BC; g_hibc = hibc ? hibc : PERIODIC_BC;
    for (int ns = 0; ns < nsweeps; ++ns) {
        if (smoother == 0) {
            for (int color = 0; color < 8; ++color) {
                nodal_fill(g, x);
                /* nodes of one colour do not depend on each other: the loop order (and the thread count) cannot change the result */
                _Pragma("omp parallel for schedule(static) num_threads(orc_threads)")
                for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
                    if (((i & 1) | ((j & 1) << 1) | ((k & 1) << 2)) != color) continue;
                    if (dm_on(i, j, k)) continue;
                    double dg, Ax = node_Ax(g, x, sig, i, j, k, &dg);
                    A4(x, i, j, k, 0) += (A4(rhs, i, j, k, 0) - Ax) / dg;
                }
            }
        } else if (smoother == 1) {
            /* lexicographic Gauss-Seidel (the reference's CPU ordering); periodic images refreshed first */
            nodal_fill(g, x);
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
                if (dm_on(i, j, k)) continue;
                double dg, Ax = node_Ax(g, x, sig, i, j, k, &dg);
                A4(x, i, j, k, 0) += (A4(rhs, i, j, k, 0) - Ax) / dg;
            }
            /* nodalSync: the duplicate (periodic image) takes the owner's value */
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
                int si = (g->periodic[0] && i == g->n[0]) ? 0 : i;
                int sj = (g->periodic[1] && j == g->n[1]) ? 0 : j;
                int sk = (g->periodic[2] && k == g->n[2]) ? 0 : k;
                A4(x, i, j, k, 0) = A4(x, si, sj, sk, 0);
            }
        } else {
            nodal_fill(g, x);
            orc_fab t = orc_alloc(g->n, ORC_NODE, 0, 1);
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
                if (dm_on(i, j, k)) { A4(&t, i, j, k, 0) = A4(x, i, j, k, 0); continue; }
                double dg, Ax = node_Ax(g, x, sig, i, j, k, &dg);
                A4(&t, i, j, k, 0) = A4(x, i, j, k, 0) + (2. / 3.) * (A4(rhs, i, j, k, 0) - Ax) / dg;
            }
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                A4(x, i, j, k, 0) = A4(&t, i, j, k, 0);
            orc_free(&t);
        }
    }
    nodal_fill(g, x);
}

/* full weighting (1,2,1)^3/64; fine must have 1 filled ghost node layer */
void orc_nodal_restrict(orc_fab* crse, const orc_fab* fine, const orc_geom* cg)
{
    for (int k = 0; k <= cg->n[2]; ++k) for (int j = 0; j <= cg->n[1]; ++j) for (int i = 0; i <= cg->n[0]; ++i) {
        int ii = 2 * i, jj = 2 * j, kk = 2 * k;
        double s = 0.0;
        for (int dk = -1; dk <= 1; ++dk) for (int dj = -1; dj <= 1; ++dj) for (int di = -1; di <= 1; ++di) {
            double w = (di == 0 ? 2. : 1.) * (dj == 0 ? 2. : 1.) * (dk == 0 ? 2. : 1.);
            s += w * A4(fine, ii + di, jj + dj, kk + dk, 0);
        }
        A4(crse, i, j, k, 0) = s * (1. / 64.);
    }
}

/* sigma-weighted (operator-dependent) interpolation, mlndlap_interpadd_aa */
static double w_side(const orc_fab* sig, int i, int j, int k, int d, int side)
{
    /* sum of the 4 fine cells adjacent to fine node (i,j,k) on the low(0)/high(1) side in direction d */
    double s = 0.0;
    for (int b = 0; b < 2; ++b) for (int a = 0; a < 2; ++a) {
        int c[3];
        int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
        int idx[3] = {i, j, k};
        c[d] = idx[d] - 1 + side; c[d1] = idx[d1] - 1 + a; c[d2] = idx[d2] - 1 + b;
        s += A4(sig, c[0], c[1], c[2], 0);
    }
    return s;
}
static double interp_line(const orc_fab* crse, const orc_fab* sig, int i, int j, int k, int ic, int jc, int kc, int d)
{
    double w1 = w_side(sig, i, j, k, d, 0), w2 = w_side(sig, i, j, k, d, 1);
    int c2[3] = {ic, jc, kc}; c2[d] += 1;
    return (A4(crse, ic, jc, kc, 0) * w1 + A4(crse, c2[0], c2[1], c2[2], 0) * w2) / (w1 + w2);
}
/* fine node in the centre of a coarse face spanned by directions d1 < d2 */
static double interp_face(const orc_fab* crse, const orc_fab* sig, int i, int j, int k, int ic, int jc, int kc, int d1, int d2)
{
    double w1 = w_side(sig, i, j, k, d1, 0), w2 = w_side(sig, i, j, k, d1, 1);
    double w3 = w_side(sig, i, j, k, d2, 0), w4 = w_side(sig, i, j, k, d2, 1);
    int f[3] = {i, j, k}, c[3] = {ic, jc, kc};
    int fm[3], fp[3], cp[3];
    double r = 0.0;
    /* neighbours along d1 are edge-centre nodes interpolated along d2, and vice versa */
    fm[0] = f[0]; fm[1] = f[1]; fm[2] = f[2]; fm[d1] -= 1;
    fp[0] = f[0]; fp[1] = f[1]; fp[2] = f[2]; fp[d1] += 1;
    cp[0] = c[0]; cp[1] = c[1]; cp[2] = c[2]; cp[d1] += 1;
    r += w1 * interp_line(crse, sig, fm[0], fm[1], fm[2], c[0], c[1], c[2], d2);
    r += w2 * interp_line(crse, sig, fp[0], fp[1], fp[2], cp[0], cp[1], cp[2], d2);
    fm[0] = f[0]; fm[1] = f[1]; fm[2] = f[2]; fm[d2] -= 1;
    fp[0] = f[0]; fp[1] = f[1]; fp[2] = f[2]; fp[d2] += 1;
    cp[0] = c[0]; cp[1] = c[1]; cp[2] = c[2]; cp[d2] += 1;
    r += w3 * interp_line(crse, sig, fm[0], fm[1], fm[2], c[0], c[1], c[2], d1);
    r += w4 * interp_line(crse, sig, fp[0], fp[1], fp[2], cp[0], cp[1], cp[2], d1);
    return r / (w1 + w2 + w3 + w4);
}

void orc_nodal_interp_add(orc_fab* fine, const orc_fab* crse, const orc_fab* sig, const orc_geom* fg)
{
    for (int k = 0; k <= fg->n[2]; ++k) for (int j = 0; j <= fg->n[1]; ++j) for (int i = 0; i <= fg->n[0]; ++i) {
        if (dm_on(i, j, k)) continue;
        int ic = i >> 1, jc = j >> 1, kc = k >> 1;
        int io = i & 1, jo = j & 1, ko = k & 1;
        double v;
        if (io && jo && ko) {
            double w[6];
            for (int d = 0; d < 3; ++d) { w[2 * d] = w_side(sig, i, j, k, d, 0); w[2 * d + 1] = w_side(sig, i, j, k, d, 1); }
            v = (w[0] * interp_face(crse, sig, i - 1, j, k, ic, jc, kc, 1, 2)
               + w[1] * interp_face(crse, sig, i + 1, j, k, ic + 1, jc, kc, 1, 2)
               + w[2] * interp_face(crse, sig, i, j - 1, k, ic, jc, kc, 0, 2)
               + w[3] * interp_face(crse, sig, i, j + 1, k, ic, jc + 1, kc, 0, 2)
               + w[4] * interp_face(crse, sig, i, j, k - 1, ic, jc, kc, 0, 1)
               + w[5] * interp_face(crse, sig, i, j, k + 1, ic, jc, kc + 1, 0, 1))
              / (w[0] + w[1] + w[2] + w[3] + w[4] + w[5]);
        } else if (jo && ko) v = interp_face(crse, sig, i, j, k, ic, jc, kc, 1, 2);
        else if (io && ko) v = interp_face(crse, sig, i, j, k, ic, jc, kc, 0, 2);
        else if (io && jo) v = interp_face(crse, sig, i, j, k, ic, jc, kc, 0, 1);
        else if (io) v = interp_line(crse, sig, i, j, k, ic, jc, kc, 0);
        else if (jo) v = interp_line(crse, sig, i, j, k, ic, jc, kc, 1);
        else if (ko) v = interp_line(crse, sig, i, j, k, ic, jc, kc, 2);
        else v = A4(crse, ic, jc, kc, 0);
        A4(fine, i, j, k, 0) += v;
    }
}

/* ------------------------------------------------------------- nodal multigrid --- */
typedef struct nlev {
    orc_geom g;
    orc_fab sig;          /* cell, 1 ghost */
    orc_fab cor, res, rescor;
    int owns_sig;
    orc_fab dm;           /* node Dirichlet mask (p == NULL: none) */
} nlev;
static inline const orc_fab* lev_dm(const nlev* L) { return L->dm.p ? &L->dm : NULL; }
static void nd_zero_masked(const orc_geom* g, orc_fab* f, const orc_fab* dm)
{
    if (!dm) return;
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
        if (A4(dm, i, j, k, 0) != 0.0) A4(f, i, j, k, 0) = 0.0;
}
/* y = A x with zero rows at Dirichlet nodes */
static void nd_adotx(const nlev* L, orc_fab* y, const orc_fab* x)
{
    orc_nodal_adotx(&L->g, y, x, &L->sig);
    nd_zero_masked(&L->g, y, lev_dm(L));
}

/* 1 on the owner copy of every node (periodic duplicates at index n excluded) */
static inline int owner(const orc_geom* g, int i, int j, int k)
{
    return !((g->periodic[0] && i == g->n[0]) || (g->periodic[1] && j == g->n[1]) || (g->periodic[2] && k == g->n[2]));
}
static double nd_norminf(const orc_geom* g, const orc_fab* f)
{
    double m = 0.0;
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
        double v = fabs(A4(f, i, j, k, 0)); if (v > m) m = v;
    }
    return m;
}
static double nd_dot(const orc_geom* g, const orc_fab* x, const orc_fab* y)
{
    double s = 0.0;
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
    { const double w = node_weight(g, g_lobc, g_hibc, i, j, k); if (w != 0.0) s += w * (A4(x, i, j, k, 0) * A4(y, i, j, k, 0)); }
    return s;
}
static void nd_subtract_mean(const orc_geom* g, orc_fab* f)
{
    double s = 0.0, cnt = 0.0;
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
    { const double w = node_weight(g, g_lobc, g_hibc, i, j, k); if (w != 0.0) { s += w * A4(f, i, j, k, 0); cnt += w; } }
    double off = s / cnt;
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
        A4(f, i, j, k, 0) -= off;
}
static void nd_sxay(const orc_geom* g, orc_fab* dst, const orc_fab* a, double s, const orc_fab* b)
{
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
        A4(dst, i, j, k, 0) = A4(a, i, j, k, 0) + s * A4(b, i, j, k, 0);
}
static void nd_copy(const orc_geom* g, orc_fab* dst, const orc_fab* src)
{
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
        A4(dst, i, j, k, 0) = A4(src, i, j, k, 0);
}
static void nd_residual(const nlev* L, orc_fab* r, orc_fab* x, const orc_fab* b)
{
    nodal_fill(&L->g, x);
    orc_nodal_adotx(&L->g, r, x, &L->sig);
    const orc_geom* g = &L->g;
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
        A4(r, i, j, k, 0) = A4(b, i, j, k, 0) - A4(r, i, j, k, 0);
    nd_zero_masked(g, r, lev_dm(L));
}

static int nd_bicgstab(const nlev* L, orc_fab* sol, const orc_fab* rhs, const orc_mg_opts* o, double eps_rel, double eps_abs, int* niters)
{
    const orc_geom* g = &L->g;
    orc_fab ph = orc_alloc(g->n, ORC_NODE, 1, 1), sh = orc_alloc(g->n, ORC_NODE, 1, 1);
    orc_fab sorig = orc_alloc(g->n, ORC_NODE, 0, 1), p = orc_alloc(g->n, ORC_NODE, 0, 1), r = orc_alloc(g->n, ORC_NODE, 0, 1);
    orc_fab s = orc_alloc(g->n, ORC_NODE, 0, 1), rh = orc_alloc(g->n, ORC_NODE, 0, 1), v = orc_alloc(g->n, ORC_NODE, 0, 1), t = orc_alloc(g->n, ORC_NODE, 0, 1);
    nd_residual(L, &r, sol, rhs);
    nd_copy(g, &sorig, sol); nd_copy(g, &rh, &r);
    orc_setval(sol, 0.0);
    double rnorm = nd_norminf(g, &r);
    const double rnorm0 = rnorm;
    int ret = 0, nit = 1;
    double rho_1 = 0, alpha = 0, omega = 0;
    if (rnorm0 == 0 || rnorm0 < eps_abs) { nit = 0; goto done; }
    /* Krylov bound: at most 2N iterations for N unique nodes (mirrors the product) */
    long nunk = 1;
    for (int d = 0; d < 3; ++d) nunk *= g->n[d] + (g->periodic[d] ? 0 : 1);
    long cap = 2 * nunk < 8 ? 8 : 2 * nunk;
    const int maxiter = (int)(o->bottom_maxiter < cap ? o->bottom_maxiter : cap);
    for (; nit <= maxiter; ++nit) {
        const double rho = nd_dot(g, &rh, &r);
        if (rho == 0) { ret = 1; break; }
        if (nit == 1) nd_copy(g, &p, &r);
        else {
            const double beta = (rho / rho_1) * (alpha / omega);
            nd_sxay(g, &p, &p, -omega, &v);
            nd_sxay(g, &p, &r, beta, &p);
        }
        nd_copy(g, &ph, &p); nodal_fill(g, &ph);
        nd_adotx(L, &v, &ph);
        const double rhTv = nd_dot(g, &rh, &v);
        if (rhTv != 0) alpha = rho / rhTv; else { ret = 2; break; }
        nd_sxay(g, sol, sol, alpha, &ph);
        nd_sxay(g, &s, &r, -alpha, &v);
        rnorm = nd_norminf(g, &s);
        if (rnorm < eps_rel * rnorm0 || rnorm < eps_abs) break;
        nd_copy(g, &sh, &s); nodal_fill(g, &sh);
        nd_adotx(L, &t, &sh);
        const double tt = nd_dot(g, &t, &t), ts = nd_dot(g, &t, &s);
        if (tt != 0) omega = ts / tt; else { ret = 3; break; }
        nd_sxay(g, sol, sol, omega, &sh);
        nd_sxay(g, &r, &s, -omega, &t);
        rnorm = nd_norminf(g, &r);
        if (rnorm < eps_rel * rnorm0 || rnorm < eps_abs) break;
        if (omega == 0) { ret = 4; break; }
        rho_1 = rho;
    }
    if (ret == 0 && rnorm > eps_rel * rnorm0 && rnorm > eps_abs) ret = 8;
    if ((ret == 0 || ret == 8) && rnorm < rnorm0) nd_sxay(g, sol, sol, 1.0, &sorig);
    else { orc_setval(sol, 0.0); nd_sxay(g, sol, sol, 1.0, &sorig); }
done:
    if (niters) *niters = nit;
    orc_free(&ph); orc_free(&sh); orc_free(&sorig); orc_free(&p); orc_free(&r); orc_free(&s); orc_free(&rh); orc_free(&v); orc_free(&t);
    return ret;
}

static void nd_smooth(const nlev* L, orc_fab* x, const orc_fab* rhs, const orc_mg_opts* o, const int lobc[3], const int hibc[3])
{
    g_dm = lev_dm(L);
    orc_nodal_smooth(&L->g, x, rhs, &L->sig, o->nodal_smoother, o->nodal_sweeps, lobc, hibc);
    g_dm = NULL;
}

static void nd_vcycle(nlev* mg, int nl, const orc_mg_opts* o, const int lobc[3], const int hibc[3], int singular, orc_mg_stats* st)
{
    for (int l = 0; l < nl - 1; ++l) {
        orc_setval(&mg[l].cor, 0.0);
        for (int i = 0; i < o->nu1; ++i) nd_smooth(&mg[l], &mg[l].cor, &mg[l].res, o, lobc, hibc);
        nd_residual(&mg[l], &mg[l].rescor, &mg[l].cor, &mg[l].res);
        nodal_fill(&mg[l].g, &mg[l].rescor);
        orc_nodal_restrict(&mg[l + 1].res, &mg[l].rescor, &mg[l + 1].g);
        nd_zero_masked(&mg[l + 1].g, &mg[l + 1].res, lev_dm(&mg[l + 1]));      /* mlndlap_restriction: Dirichlet coarse nodes get 0 */
    }
    {
        nlev* b = &mg[nl - 1];
        orc_setval(&b->cor, 0.0);
        if (o->bottom_smoother_only) {
            for (int i = 0; i < o->nuf; ++i) nd_smooth(b, &b->cor, &b->res, o, lobc, hibc);
        } else {
            orc_fab rb = orc_alloc(b->g.n, ORC_NODE, 0, 1);
            nd_copy(&b->g, &rb, &b->res);
            if (singular) nd_subtract_mean(&b->g, &rb);
            int nit = 0;
            int ret = nd_bicgstab(b, &b->cor, &rb, o, o->bottom_reltol, -1.0, &nit);
            if (st) st->bottom_iters_total += nit;
            if (ret != 0) {
                orc_setval(&b->cor, 0.0);
                for (int i = 0; i < o->nuf; ++i) nd_smooth(b, &b->cor, &b->res, o, lobc, hibc);
            }
            int nn = ret == 0 ? o->nub : o->nuf;
            for (int i = 0; i < nn; ++i) nd_smooth(b, &b->cor, &b->res, o, lobc, hibc);
            orc_free(&rb);
        }
    }
    for (int l = nl - 2; l >= 0; --l) {
        nodal_fill(&mg[l + 1].g, &mg[l + 1].cor);
        g_dm = lev_dm(&mg[l]);
        orc_nodal_interp_add(&mg[l].cor, &mg[l + 1].cor, &mg[l].sig, &mg[l].g);
        g_dm = NULL;
        for (int i = 0; i < o->nu2; ++i) nd_smooth(&mg[l], &mg[l].cor, &mg[l].res, o, lobc, hibc);
    }
}

/* Dirichlet node mask of one MG level: a node is Dirichlet if one of the 8 cells around it is not part of the problem --
 * outside a Dirichlet (outflow) domain face, or not covered by the level (cov: cell fab, != 0 on covered cells; NULL: the
 * level covers the domain).  Cells beyond a periodic face are the periodic images, cells beyond a Neumann wall mirror the
 * cells inside. */
static int cell_in(const orc_geom* g, const int lobc[3], const int hibc[3], const orc_fab* cov, int ci, int cj, int ck)
{
    int c[3] = {ci, cj, ck};
    for (int d = 0; d < 3; ++d) {
        if (c[d] < 0) {
            if (g->periodic[d]) c[d] += g->n[d];
            else if (NEU(lobc[d])) c[d] = -c[d] - 1;
            else return 0;
        } else if (c[d] >= g->n[d]) {
            if (g->periodic[d]) c[d] -= g->n[d];
            else if (NEU(hibc[d])) c[d] = 2 * g->n[d] - 1 - c[d];
            else return 0;
        }
    }
    return cov ? A4(cov, c[0], c[1], c[2], 0) != 0.0 : 1;
}
static int build_dmask(const orc_geom* g, const int lobc[3], const int hibc[3], const orc_fab* cov, orc_fab* dm)
{
    int any = 0;
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
        int in = 1;
        for (int c = 0; c < 8 && in; ++c) in = cell_in(g, lobc, hibc, cov, i - 1 + (c & 1), j - 1 + ((c >> 1) & 1), k - 1 + ((c >> 2) & 1));
        A4(dm, i, j, k, 0) = in ? 0.0 : 1.0;
        any |= !in;
    }
    return any;
}

void orc_nodal_solve(const orc_geom* g, orc_fab* phi, const orc_fab* rhs_in, const orc_fab* sig,
                     const int lobc[3], const int hibc[3], double rtol, double atol,
                     const orc_mg_opts* o, orc_mg_stats* st)
{
    orc_nodal_solve_cov(g, phi, rhs_in, sig, lobc, hibc, NULL, rtol, atol, o, st);
}

/* cov (optional): cell fab, != 0 on the cells of the level.  With cov or Dirichlet faces the nodes on the boundary of the
 * covered region hold Dirichlet data (phi keeps its incoming value there); sigma is taken as 0 on uncovered cells. */
void orc_nodal_solve_cov(const orc_geom* g, orc_fab* phi, const orc_fab* rhs_in, const orc_fab* sig,
                         const int lobc[3], const int hibc[3], const orc_fab* cov, double rtol, double atol,
                         const orc_mg_opts* o, orc_mg_stats* st)
{
    nlev mg[32];
    memset(mg, 0, sizeof(mg));
    g_lobc = lobc ? lobc : PERIODIC_BC; g_hibc = hibc ? hibc : PERIODIC_BC;
    int nl = 1;
    mg[0].g = *g;
    mg[0].sig = orc_alloc(g->n, ORC_CELL, 1, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        A4(&mg[0].sig, i, j, k, 0) = (cov && A4(cov, i, j, k, 0) == 0.0) ? 0.0 : A4(sig, i, j, k, 0);
    sigma_fill_bc(g, &mg[0].sig);
    orc_fab covl[32];
    memset(covl, 0, sizeof(covl));
    if (cov) {
        covl[0] = orc_alloc(g->n, ORC_CELL, 0, 1);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
            A4(&covl[0], i, j, k, 0) = A4(cov, i, j, k, 0) != 0.0 ? 1.0 : 0.0;
    }
    while (nl <= o->max_coarsening_level && nl < 32) {
        const orc_geom* fg = &mg[nl - 1].g;
        int ok = 1;
        for (int d = 0; d < 3; ++d) if (fg->n[d] % 2 != 0 || fg->n[d] / 2 < o->min_width) ok = 0;
        if (!ok) break;
        mg[nl].g = *fg;
        for (int d = 0; d < 3; ++d) { mg[nl].g.n[d] = fg->n[d] / 2; mg[nl].g.dx[d] = fg->dx[d] * 2.0; }
        mg[nl].sig = orc_alloc(mg[nl].g.n, ORC_CELL, 1, 1);
        orc_cc_restrict(&mg[nl].sig, &mg[nl - 1].sig, mg[nl].g.n);
        sigma_fill_bc(&mg[nl].g, &mg[nl].sig);
        if (cov) {
            /* the level's boxes coarsen with the multigrid (aligned to 2^l by construction): a coarse cell is covered if its
             * first fine cell is */
            covl[nl] = orc_alloc(mg[nl].g.n, ORC_CELL, 0, 1);
            for (int k = 0; k < mg[nl].g.n[2]; ++k) for (int j = 0; j < mg[nl].g.n[1]; ++j) for (int i = 0; i < mg[nl].g.n[0]; ++i)
                A4(&covl[nl], i, j, k, 0) = A4(&covl[nl - 1], 2 * i, 2 * j, 2 * k, 0);
        }
        ++nl;
    }
    int singular = 1;
    for (int l = 0; l < nl; ++l) {
        mg[l].dm = orc_alloc(mg[l].g.n, ORC_NODE, 0, 1);
        if (!build_dmask(&mg[l].g, g_lobc, g_hibc, cov ? &covl[l] : NULL, &mg[l].dm)) orc_free(&mg[l].dm);
        else singular = 0;
    }
    for (int l = 0; l < nl; ++l) {
        mg[l].cor = orc_alloc(mg[l].g.n, ORC_NODE, 1, 1);
        mg[l].res = orc_alloc(mg[l].g.n, ORC_NODE, 1, 1);
        mg[l].rescor = orc_alloc(mg[l].g.n, ORC_NODE, 1, 1);
    }
    orc_mg_stats loc; memset(&loc, 0, sizeof(loc));

    orc_fab rhs = orc_alloc(g->n, ORC_NODE, 0, 1);
    nd_copy(g, &rhs, rhs_in);
    if (singular) nd_subtract_mean(g, &rhs);
    orc_fab* res = &mg[0].res;
    nd_residual(&mg[0], res, phi, &rhs);
    loc.resnorm0 = nd_norminf(g, res);
    loc.rhsnorm0 = nd_norminf(g, &rhs);
    const double max_norm = loc.rhsnorm0 >= loc.resnorm0 ? loc.rhsnorm0 : loc.resnorm0;
    const double res_target = fmax(atol, fmax(rtol, 1.e-16) * max_norm);
    loc.resnorm = loc.resnorm0;
    if (o->verbose) printf("orc nodal MLMG: rhs %.6e resid0 %.6e levels %d\n", loc.rhsnorm0, loc.resnorm0, nl);
    if (o->fixed_iters <= 0 && loc.resnorm0 <= res_target) loc.converged = 1;
    else {
        const int maxit = o->fixed_iters > 0 ? o->fixed_iters : o->max_iters;
        for (int iter = 0; iter < maxit; ++iter) {
            if (singular) nd_subtract_mean(g, res);
            nd_vcycle(mg, nl, o, lobc, hibc, singular, &loc);
            nd_sxay(g, phi, phi, 1.0, &mg[0].cor);
            nd_residual(&mg[0], res, phi, &rhs);
            loc.resnorm = nd_norminf(g, res);
            loc.iters = iter + 1;
            if (o->verbose) printf("orc nodal MLMG: iter %d resid %.6e\n", iter + 1, loc.resnorm);
            if (o->fixed_iters <= 0 && loc.resnorm <= res_target) { loc.converged = 1; break; }
        }
    }
    nodal_fill(g, phi);
    if (st) *st = loc;
    orc_free(&rhs);
    for (int l = 0; l < nl; ++l) {
        orc_free(&mg[l].cor); orc_free(&mg[l].res); orc_free(&mg[l].rescor); orc_free(&mg[l].sig);
        if (mg[l].dm.p) orc_free(&mg[l].dm);
        if (covl[l].p) orc_free(&covl[l]);
    }
}

void orc_nodal_project(const orc_geom* g, orc_fab* vel, orc_fab* phi, const orc_fab* sig,
                       const int lobc[3], const int hibc[3], double rtol, double atol,
                       const orc_mg_opts* o, orc_mg_stats* st)
{
    orc_fab rhs = orc_alloc(g->n, ORC_NODE, 0, 1);
    orc_nodal_divu_bc(g, &rhs, vel, lobc, hibc);
    orc_nodal_solve(g, phi, &rhs, sig, lobc, hibc, rtol, atol, o, st);
    orc_nodal_mknewu(g, vel, phi, sig);
    orc_free(&rhs);
}

/* Projection::doMLMGNodalProjection on one AMR level > 0 (Source/Projection.cpp:2385-2567 with c_lev > 0, nlevel = 1): the level
 * covers the cells with cov != 0; the nodes on its boundary inside the domain keep the incoming phi (Dirichlet mask).  rhs = div(vel)
 * needs vel on the level only (boundary nodes carry no equation); vel -= sig grad phi on the level's cells. */
void orc_nodal_project_cov(const orc_geom* g, orc_fab* vel, orc_fab* phi, const orc_fab* sig, const int lobc[3], const int hibc[3],
                           const orc_fab* cov, double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st)
{
    orc_fab rhs = orc_alloc(g->n, ORC_NODE, 0, 1);
    orc_nodal_divu_bc(g, &rhs, vel, lobc, hibc);
    orc_nodal_solve_cov(g, phi, &rhs, sig, lobc, hibc, cov, rtol, atol, o, st);
    orc_fab v2 = orc_alloc(g->n, ORC_CELL, 0, 3);
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&v2, i, j, k, n) = A4(vel, i, j, k, n);
    orc_nodal_mknewu(g, &v2, phi, sig);
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
        if (!cov || A4(cov, i, j, k, 0) != 0.0) A4(vel, i, j, k, n) = A4(&v2, i, j, k, n);
    orc_free(&v2); orc_free(&rhs);
}

/* MLNodeLaplacian::compRHS, the cell-centred source (mlndlap_rhcc, then mlndlap_impose_neumann_bc on the sum with div(vel)):
 * rhs(node) += 1/8 of the sum over the 8 cells around the node -- cells outside a non-periodic domain face, or outside `cov`
 * (NULL: the whole domain), count zero -- doubled per Neumann / inflow wall direction the node lies on */
void orc_nodal_rhcc_add(const orc_geom* g, orc_fab* rhs, const orc_fab* rhcc, const int lobc[3], const int hibc[3], const orc_fab* cov)
{
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
        double s = 0.0;
        for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
            int cell[3] = {i - 1 + cx, j - 1 + cy, k - 1 + cz}, out = 0;
            for (int e = 0; e < 3; ++e) {
                if (cell[e] >= 0 && cell[e] < g->n[e]) continue;
                if (g->periodic[e]) cell[e] = (cell[e] % g->n[e] + g->n[e]) % g->n[e]; else out = 1;
            }
            if (out) continue;
            if (cov && A4(cov, cell[0], cell[1], cell[2], 0) == 0.0) continue;
            s += A4(rhcc, cell[0], cell[1], cell[2], 0);
        }
        double r = 0.125 * s;
        const int idx[3] = {i, j, k};
        for (int e = 0; e < 3; ++e) {
            if (g->periodic[e]) continue;
            if (idx[e] == 0 && NEU(lobc[e])) r *= 2.0;
            if (idx[e] == g->n[e] && NEU(hibc[e])) r *= 2.0;
        }
        A4(rhs, i, j, k, 0) += r;
    }
}

/* orc_nodal_project / orc_nodal_project_cov with a cell-centred source: div(sig grad phi) = div(vel) + <rhcc> (Hydro::NodalProjector
 * with rhcc; Projection.cpp passes rhcc = -divu/dt) */
void orc_nodal_project_rhcc(const orc_geom* g, orc_fab* vel, orc_fab* phi, const orc_fab* sig, const int lobc[3], const int hibc[3],
                            const orc_fab* cov, const orc_fab* rhcc, double rtol, double atol, const orc_mg_opts* o, orc_mg_stats* st)
{
    orc_fab rhs = orc_alloc(g->n, ORC_NODE, 0, 1);
    orc_nodal_divu_bc(g, &rhs, vel, lobc, hibc);
    if (rhcc) orc_nodal_rhcc_add(g, &rhs, rhcc, lobc, hibc, cov);
    if (!cov) { orc_nodal_solve(g, phi, &rhs, sig, lobc, hibc, rtol, atol, o, st); orc_nodal_mknewu(g, vel, phi, sig); }
    else {
        orc_nodal_solve_cov(g, phi, &rhs, sig, lobc, hibc, cov, rtol, atol, o, st);
        orc_fab v2 = orc_alloc(g->n, ORC_CELL, 0, 3);
        for (int n = 0; n < 3; ++n)
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&v2, i, j, k, n) = A4(vel, i, j, k, n);
        orc_nodal_mknewu(g, &v2, phi, sig);
        for (int n = 0; n < 3; ++n)
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
            if (A4(cov, i, j, k, 0) != 0.0) A4(vel, i, j, k, n) = A4(&v2, i, j, k, n);
        orc_free(&v2);
    }
    orc_free(&rhs);
}
