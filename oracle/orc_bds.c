/* oracle/orc_bds.c -- the Bell-Dawson-Shubin (BDS) edge states (ns.advection_scheme = BDS, Source/NavierStokesBase.cpp:548-553; the
 * ComputeFluxesOnBoxFromState(..., "BDS") call of NavierStokesBase::ComputeAofs, :4701-4717) restated on the CPU.  TEST INFRASTRUCTURE
 * ONLY (see orc.h).  PARITY UNPINNED: the scheme lives in AMReX-Hydro (hydro_bds_edge_state_3D.cpp, BDS::ComputeEdgeState), which is not
 * in the reference tree; it is restated from the published algorithm the reference's documentation points to
 * (Docs/sphinx_documentation/source/TimeStep.rst:92-133; Nonaka, May, Almgren, Bell, SIAM J. Sci. Comput. 33 (2011) 2039-2062):
 *
 *  Step 1 (slopes).  Fourth-order interpolation of the cell averages to the nodes (tensor product of (-1, 7, 7, -1)/12); nodes on or
 *    beyond a physical boundary take the average of the adjacent ghost cells (which hold the boundary value).  From the 8 corner values
 *    of a cell: s_x, s_y, s_z, s_xy, s_xz, s_yz, s_xyz.  Limiting: the trilinear polynomial evaluated at the 8 corners is clipped to
 *    the min / max of the 8 cells sharing each corner, then three passes redistribute the resulting change of the cell mean over the
 *    corners that still have room, and the slopes are recomputed from the corners.
 *  Step 2 (edge states).  For a face of direction D with normal velocity u: upwind cell I; s = polynomial at the centroid of the swept
 *    slab, x (1 - dt/2 u_x) [conservative] or x (1 + dt/2 (v_y + w_z)) [convective], + dt/2 f; minus the transverse flux differences
 *    dt/(2 h_T) (Gamma^{T+} v^{T+} - Gamma^{T-} v^{T-}) for both transverse directions T.  Gamma^{T+-}: the polynomial of the cell upwind
 *    of the transverse face averaged over the triangle (p1, p2, p3) of the traced region (mid-point rule), with its own source factor,
 *    minus dt/(3 h_O) (Gamma^{T,O+} w^{O+} - Gamma^{T,O-} w^{O-}) for the remaining direction O, where Gamma^{T,O+-} is the average
 *    over the tetrahedron (p1 .. p4) by the five-point rule (-4/5 at the centroid, 9/20 at the points (1/2, 1/6, 1/6, 1/6)).
 *    A velocity taken from a neighbouring face enters a traced point only if it has the sign of the face's own velocity.
 *  On a face of the physical boundary the edge state is the ghost-cell value (the boundary value).
 * The periodic form is pinned by known answers (tests/test_cpu_bds.py): exact for trilinear data under a constant velocity field,
 * conservative, second-order accurate, bounded for a step profile. */
#include "orc_int.h"

static int g_scheme = 0;
void orc_godunov_set_scheme(int scheme) { g_scheme = scheme; }
int orc_godunov_get_scheme(void) { return g_scheme; }

static inline int is_phys(int b) { return b == ORC_BC_FOEXTRAP || b == ORC_BC_HOEXTRAP || b == ORC_BC_EXT_DIR; }

static inline double bds_eval(double s, const double* sl, const double* del)
{
    return s + del[0] * sl[0] + del[1] * sl[1] + del[2] * sl[2] + del[0] * del[1] * sl[3] + del[0] * del[2] * sl[4] + del[1] * del[2] * sl[5]
             + del[0] * del[1] * del[2] * sl[6];
}

/* slopes (7 comps) on the cells of the domain grown by one, for component n of s (>= 3 filled ghost cells) */
static orc_fab bds_slopes(const orc_geom* g, const orc_fab* s, int n, const orc_bcrec* bc)
{
    const double h[3] = {g->dx[0], g->dx[1], g->dx[2]};
    const int nn[3] = {g->n[0], g->n[1], g->n[2]};
    orc_fab sint = orc_alloc(nn, ORC_NODE, 2, 1);        /* nodes -2 .. n+2 (needed: -1 .. n+1) */
    orc_fab sl = orc_alloc(nn, ORC_CELL, 1, 7);
    static const double w4[4] = {-1.0 / 12.0, 7.0 / 12.0, 7.0 / 12.0, -1.0 / 12.0};
    int lo_phys[3], hi_phys[3];
    for (int d = 0; d < 3; ++d) {
        lo_phys[d] = !g->periodic[d] && bc && is_phys(bc->lo[d]);
        hi_phys[d] = !g->periodic[d] && bc && is_phys(bc->hi[d]);
    }
    for (int k = -1; k <= nn[2] + 1; ++k) for (int j = -1; j <= nn[1] + 1; ++j) for (int i = -1; i <= nn[0] + 1; ++i) {
        const int idx[3] = {i, j, k};
        int bd = -1, bcell = 0;
        for (int d = 0; d < 3 && bd < 0; ++d) {
            if (lo_phys[d] && idx[d] <= 0) { bd = d; bcell = -1; }
            else if (hi_phys[d] && idx[d] >= nn[d]) { bd = d; bcell = nn[d]; }
        }
        double v = 0.0;
        if (bd >= 0) {
            /* node on / beyond a physical boundary: the average of the four ghost cells around it in the boundary plane */
            const int t1 = (bd + 1) % 3, t2 = (bd + 2) % 3;
            for (int b = -1; b <= 0; ++b) for (int a = -1; a <= 0; ++a) {
                int c[3];
                c[bd] = bcell; c[t1] = idx[t1] + a; c[t2] = idx[t2] + b;
                v += 0.25 * A4(s, c[0], c[1], c[2], n);
            }
        } else {
            for (int c = 0; c < 4; ++c) for (int b = 0; b < 4; ++b) for (int a = 0; a < 4; ++a)
                v += w4[a] * w4[b] * w4[c] * A4(s, i - 2 + a, j - 2 + b, k - 2 + c, n);
        }
        A4(&sint, i, j, k, 0) = v;
    }
    const double eps = 1.0e-10;
    for (int k = -1; k <= nn[2]; ++k) for (int j = -1; j <= nn[1]; ++j) for (int i = -1; i <= nn[0]; ++i) {
        /* corner m = (mx, my, mz) in {0,1}^3 <-> node (i + mx, j + my, k + mz) */
        double sc[8], smin[8], smax[8];
        const double s0 = A4(s, i, j, k, n);
        for (int m = 0; m < 8; ++m) sc[m] = A4(&sint, i + (m & 1), j + ((m >> 1) & 1), k + ((m >> 2) & 1), 0);
        double sl7[7];
#define CORNERS_TO_SLOPES()                                                                                                        \
        do {                                                                                                                       \
            sl7[0] = 0.25 * ((sc[1] + sc[3] + sc[5] + sc[7]) - (sc[0] + sc[2] + sc[4] + sc[6])) / h[0];                            \
            sl7[1] = 0.25 * ((sc[2] + sc[3] + sc[6] + sc[7]) - (sc[0] + sc[1] + sc[4] + sc[5])) / h[1];                            \
            sl7[2] = 0.25 * ((sc[4] + sc[5] + sc[6] + sc[7]) - (sc[0] + sc[1] + sc[2] + sc[3])) / h[2];                            \
            sl7[3] = 0.5 * ((sc[0] + sc[3] + sc[4] + sc[7]) - (sc[1] + sc[2] + sc[5] + sc[6])) / (h[0] * h[1]);                    \
            sl7[4] = 0.5 * ((sc[0] + sc[5] + sc[2] + sc[7]) - (sc[1] + sc[4] + sc[3] + sc[6])) / (h[0] * h[2]);                    \
            sl7[5] = 0.5 * ((sc[0] + sc[6] + sc[1] + sc[7]) - (sc[2] + sc[4] + sc[3] + sc[5])) / (h[1] * h[2]);                    \
            sl7[6] = ((sc[7] + sc[1] + sc[2] + sc[4]) - (sc[0] + sc[3] + sc[5] + sc[6])) / (h[0] * h[1] * h[2]);                   \
        } while (0)
        CORNERS_TO_SLOPES();
        /* the polynomial at the corners, clipped to the bounds of the 8 cells around each corner */
        for (int m = 0; m < 8; ++m) {
            const int mx = m & 1, my = (m >> 1) & 1, mz = (m >> 2) & 1;
            const double del[3] = {(mx ? 0.5 : -0.5) * h[0], (my ? 0.5 : -0.5) * h[1], (mz ? 0.5 : -0.5) * h[2]};
            sc[m] = bds_eval(s0, sl7, del);
            double mn = s0, mxv = s0;
            for (int c = -1; c <= 0; ++c) for (int b = -1; b <= 0; ++b) for (int a = -1; a <= 0; ++a) {
                const double q = A4(s, i + mx + a, j + my + b, k + mz + c, n);
                mn = fmin(mn, q); mxv = fmax(mxv, q);
            }
            smin[m] = mn; smax[m] = mxv;
            sc[m] = fmax(fmin(sc[m], smax[m]), smin[m]);
        }
        for (int ll = 0; ll < 3; ++ll) {
            double sumloc = 0.0;
            for (int m = 0; m < 8; ++m) sumloc += sc[m];
            sumloc *= 0.125;
            double sumdif = (sumloc - s0) * 8.0;
            const double sgndif = copysign(1.0, sumdif);
            double diff[8];
            int kdp = 0;
            for (int m = 0; m < 8; ++m) { diff[m] = (sc[m] - s0) * sgndif; if (diff[m] > eps) ++kdp; }
            for (int m = 0; m < 8; ++m) {
                const double div = kdp < 1 ? 1.0 : (double)kdp;
                double redfac = 0.0;
                if (diff[m] > eps) { redfac = sumdif * sgndif / div; --kdp; }
                const double redmax = sgndif > 0.0 ? sc[m] - smin[m] : smax[m] - sc[m];
                redfac = fmin(redfac, redmax);
                sumdif -= redfac * sgndif;
                sc[m] -= redfac * sgndif;
            }
        }
        CORNERS_TO_SLOPES();
#undef CORNERS_TO_SLOPES
        for (int q = 0; q < 7; ++q) A4(&sl, i, j, k, q) = sl7[q];
    }
    orc_free(&sint);
    return sl;
}

typedef struct {
    const orc_geom* g; const orc_fab* s; int n; const orc_fab* sl; orc_fab* const* mac; int conserv; double dt;
} bds_ctx;

static inline double mac_at(const bds_ctx* c, int d, const int f[3]) { return A4(c->mac[d], f[0], f[1], f[2], 0); }
/* d(u_d)/dx_d in cell q */
static inline double dvel(const bds_ctx* c, int d, const int q[3])
{
    int p[3] = {q[0], q[1], q[2]};
    p[d] += 1;
    return (mac_at(c, d, p) - mac_at(c, d, q)) / c->g->dx[d];
}
static inline double poly(const bds_ctx* c, const int q[3], const double del[3])
{
    double sl7[7];
    for (int m = 0; m < 7; ++m) sl7[m] = A4(c->sl, q[0], q[1], q[2], m);
    return bds_eval(A4(c->s, q[0], q[1], q[2], c->n), sl7, del);
}

/* edge state on face f of direction D */
static double bds_edge(const bds_ctx* c, int D, const int f[3], const orc_fab* fq)
{
    const double dt = c->dt, dt2 = dt / 2.0, dt3 = dt / 3.0, dt4 = dt / 4.0;
    const double* h = c->g->dx;
    const double uD = mac_at(c, D, f);
    const int sgnD = uD > 0.0 ? 1 : -1, offD = uD > 0.0 ? -1 : 0;
    int I[3] = {f[0], f[1], f[2]};
    I[D] += offD;
    double del[3] = {0.0, 0.0, 0.0};
    del[D] = sgnD * 0.5 * h[D] - 0.5 * uD * dt;
    double sedge = poly(c, I, del);
    const int Ta = (D + 1) % 3, Tb = (D + 2) % 3;
    if (c->conserv) sedge *= 1.0 - dt2 * dvel(c, D, I);
    else sedge *= 1.0 + dt2 * (dvel(c, Ta, I) + dvel(c, Tb, I));
    if (fq) sedge += dt2 * A4(fq, I[0], I[1], I[2], c->n);
    for (int pass = 0; pass < 2; ++pass) {
        const int T = pass == 0 ? Ta : Tb, O = pass == 0 ? Tb : Ta;
        for (int sideT = 1; sideT >= -1; sideT -= 2) {
            int ft[3] = {I[0], I[1], I[2]};
            if (sideT > 0) ft[T] += 1;
            const double V = mac_at(c, T, ft);
            const int sgnT = V > 0.0 ? 1 : -1;
            const int offT = sideT > 0 ? (V > 0.0 ? 0 : 1) : (V > 0.0 ? -1 : 0);
            int J[3] = {I[0], I[1], I[2]};
            J[T] += offT;
            int fD[3] = {f[0], f[1], f[2]};
            fD[T] += offT;
            const double uS = mac_at(c, D, fD);
            const double u = uD * uS > 0.0 ? uS : 0.0;
            double p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0}, p3[3] = {0, 0, 0};
            p1[D] = sgnD * 0.5 * h[D];           p1[T] = sgnT * 0.5 * h[T];
            p2[D] = sgnD * 0.5 * h[D] - uD * dt; p2[T] = sgnT * 0.5 * h[T];
            p3[D] = sgnD * 0.5 * h[D] - u * dt;  p3[T] = sgnT * 0.5 * h[T] - V * dt;
            double d1[3], d2[3], d3[3];
            for (int l = 0; l < 3; ++l) { d1[l] = 0.5 * (p2[l] + p3[l]); d2[l] = 0.5 * (p1[l] + p3[l]); d3[l] = 0.5 * (p1[l] + p2[l]); }
            double gamma = (poly(c, J, d1) + poly(c, J, d2) + poly(c, J, d3)) / 3.0;
            if (c->conserv) gamma *= 1.0 - dt3 * (dvel(c, D, J) + dvel(c, T, J));
            else gamma *= 1.0 + dt3 * dvel(c, O, J);
            for (int sideO = 1; sideO >= -1; sideO -= 2) {
                int fo[3] = {J[0], J[1], J[2]};
                if (sideO > 0) fo[O] += 1;
                const double W = mac_at(c, O, fo);
                const int sgnO = W > 0.0 ? 1 : -1;
                const int offO = sideO > 0 ? (W > 0.0 ? 0 : 1) : (W > 0.0 ? -1 : 0);
                int K[3] = {J[0], J[1], J[2]};
                K[O] += offO;
                int fD2[3] = {fD[0], fD[1], fD[2]};
                fD2[O] += offO;
                const double uS2 = mac_at(c, D, fD2);
                const double uu = uD * uS2 > 0.0 ? uS2 : 0.0;
                int ft2[3] = {ft[0], ft[1], ft[2]};
                ft2[O] += offO;
                const double vS2 = mac_at(c, T, ft2);
                const double vv = V * vS2 > 0.0 ? vS2 : 0.0;
                double q1[3], q2[3], q3[3], q4[3];
                for (int l = 0; l < 3; ++l) { q1[l] = p1[l]; q2[l] = p2[l]; q3[l] = p3[l]; q4[l] = 0.0; }
                q1[O] = q2[O] = q3[O] = sgnO * 0.5 * h[O];
                q4[D] = sgnD * 0.5 * h[D] - uu * dt; q4[T] = sgnT * 0.5 * h[T] - vv * dt; q4[O] = sgnO * 0.5 * h[O] - W * dt;
                double e1[3], e2[3], e3[3], e4[3], e5[3];
                const double a = 0.5, b = 1.0 / 6.0;
                for (int l = 0; l < 3; ++l) {
                    e1[l] = a * q1[l] + b * q2[l] + b * q3[l] + b * q4[l];
                    e2[l] = b * q1[l] + a * q2[l] + b * q3[l] + b * q4[l];
                    e3[l] = b * q1[l] + b * q2[l] + a * q3[l] + b * q4[l];
                    e4[l] = b * q1[l] + b * q2[l] + b * q3[l] + a * q4[l];
                    e5[l] = 0.25 * (q1[l] + q2[l] + q3[l] + q4[l]);
                }
                double gamma2 = -0.8 * poly(c, K, e5) + 0.45 * (poly(c, K, e1) + poly(c, K, e2) + poly(c, K, e3) + poly(c, K, e4));
                if (c->conserv) gamma2 *= 1.0 - dt4 * (dvel(c, D, K) + dvel(c, T, K) + dvel(c, O, K));
                gamma2 *= W;
                gamma -= (double)sideO * dt * gamma2 / (3.0 * h[O]);
            }
            gamma *= V;
            sedge -= (double)sideT * dt * gamma / (2.0 * h[T]);
        }
    }
    return sedge;
}

/* BDS::ComputeEdgeState: edge[d] (ncomp face comps, the valid faces of the whole-domain arrays).  q: >= 3 filled ghost cells (physical BCs
 * included), mac: >= 1 filled ghost layer, fq: the forcing (may be NULL) */
void orc_bds_edge_state(const orc_geom* g, const orc_fab* q, int ncomp, const orc_fab* fq, orc_fab* const mac[3], const int* iconserv,
                        double dt, const orc_bcrec* bc, int is_velocity, orc_fab* edge)
{
    for (int n = 0; n < ncomp; ++n) {
        orc_fab sl = bds_slopes(g, q, n, bc ? &bc[n] : NULL);
        bds_ctx c = {g, q, n, &sl, mac, iconserv[n], dt};
        for (int D = 0; D < 3; ++D) {
            const int lo_p = !g->periodic[D] && bc && is_phys(bc[n].lo[D]), hi_p = !g->periodic[D] && bc && is_phys(bc[n].hi[D]);
            for (int k = 0; k <= g->n[2] - (D == 2 ? 0 : 1); ++k) for (int j = 0; j <= g->n[1] - (D == 1 ? 0 : 1); ++j)
            for (int i = 0; i <= g->n[0] - (D == 0 ? 0 : 1); ++i) {
                const int f[3] = {i, j, k};
                double v;
                if (lo_p && f[D] == 0) { int cc[3] = {i, j, k}; cc[D] = -1; v = A4(q, cc[0], cc[1], cc[2], n); }
                else if (hi_p && f[D] == g->n[D]) v = A4(q, i, j, k, n);
                else v = bds_edge(&c, D, f, fq);
                (void)is_velocity;
                A4(&edge[D], i, j, k, n) = v;
            }
        }
        orc_free(&sl);
    }
}
