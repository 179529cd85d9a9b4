/* oracle/orc_fill.c -- ghost-cell filling for the CPU oracle (test infrastructure only;
 * see orc.h header: PARITY UNPINNED).
 *
 * orc_fill_periodic  : amrex FillBoundary(geom.periodicity()) on a single-box level
 *                      (reference call sites: Source/MacProj.cpp:1127, Source/Projection.cpp:338-339,
 *                      and inside every FillPatch).
 * orc_fill_physbc_cc : cell-centred physical-BC fill (BCType semantics, reference
 *                      Source/NS_BC.H:7-55 tables + Source/NS_bcfill.H:17-95 ext_dir functors).
 */
#include "orc_int.h"

static inline int wrap(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }

void orc_fill_periodic(orc_fab* f, const orc_geom* g, const int type[3])
{
    int vhi[3];
    for (int d = 0; d < 3; ++d) vhi[d] = g->n[d] - 1 + type[d];
    for (int n = 0; n < f->nc; ++n)
    for (int k = f->lo[2]; k <= f->hi[2]; ++k)
    for (int j = f->lo[1]; j <= f->hi[1]; ++j)
    for (int i = f->lo[0]; i <= f->hi[0]; ++i) {
        int idx[3] = {i, j, k}, src[3] = {i, j, k};
        int outside = 0, ok = 1;
        for (int d = 0; d < 3; ++d) {
            if (idx[d] < 0 || idx[d] > vhi[d]) {
                outside = 1;
                if (g->periodic[d]) src[d] = wrap(idx[d], g->n[d]);
                else ok = 0;
            }
        }
        if (outside && ok) A4(f, i, j, k, n) = A4(f, src[0], src[1], src[2], n);
    }
}

/* one direction of the cell-centred BC fill */
static void physbc_dir(orc_fab* f, const orc_geom* g, int d, int n, int bclo, int bchi,
                       double edlo, double edhi)
{
    const int dlo = 0, dhi = g->n[d] - 1;
    int lo[3] = {f->lo[0], f->lo[1], f->lo[2]}, hi[3] = {f->hi[0], f->hi[1], f->hi[2]};
    for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
    for (int i = lo[0]; i <= hi[0]; ++i) {
        int idx[3] = {i, j, k};
        int c = idx[d];
        if (c >= dlo && c <= dhi) continue;
        int s[3] = {i, j, k};
        double v;
        if (c < dlo) {
            switch (bclo) {
            case ORC_BC_FOEXTRAP: s[d] = dlo; v = A4(f, s[0], s[1], s[2], n); break;
            case ORC_BC_HOEXTRAP:
                if (c < dlo - 1) { s[d] = dlo; v = A4(f, s[0], s[1], s[2], n); }
                else {
                    int s1[3] = {i, j, k}, s2[3] = {i, j, k}, s3[3] = {i, j, k};
                    s1[d] = dlo; s2[d] = dlo + 1; s3[d] = dlo + 2;
                    if (dlo + 2 <= (f->hi[d] < dhi ? f->hi[d] : dhi))
                        v = 0.125 * (15. * A4(f, s1[0], s1[1], s1[2], n) - 10. * A4(f, s2[0], s2[1], s2[2], n) + 3. * A4(f, s3[0], s3[1], s3[2], n));
                    else
                        v = 0.5 * (3. * A4(f, s1[0], s1[1], s1[2], n) - A4(f, s2[0], s2[1], s2[2], n));
                }
                break;
            case ORC_BC_REFLECT_EVEN: s[d] = 2 * dlo - c - 1; v = A4(f, s[0], s[1], s[2], n); break;
            case ORC_BC_REFLECT_ODD: s[d] = 2 * dlo - c - 1; v = -A4(f, s[0], s[1], s[2], n); break;
            case ORC_BC_EXT_DIR: v = edlo; break;
            default: continue;
            }
        } else {
            switch (bchi) {
            case ORC_BC_FOEXTRAP: s[d] = dhi; v = A4(f, s[0], s[1], s[2], n); break;
            case ORC_BC_HOEXTRAP:
                if (c > dhi + 1) { s[d] = dhi; v = A4(f, s[0], s[1], s[2], n); }
                else {
                    int s1[3] = {i, j, k}, s2[3] = {i, j, k}, s3[3] = {i, j, k};
                    s1[d] = dhi; s2[d] = dhi - 1; s3[d] = dhi - 2;
                    if (dhi - 2 >= (f->lo[d] > dlo ? f->lo[d] : dlo))
                        v = 0.125 * (15. * A4(f, s1[0], s1[1], s1[2], n) - 10. * A4(f, s2[0], s2[1], s2[2], n) + 3. * A4(f, s3[0], s3[1], s3[2], n));
                    else
                        v = 0.5 * (3. * A4(f, s1[0], s1[1], s1[2], n) - A4(f, s2[0], s2[1], s2[2], n));
                }
                break;
            case ORC_BC_REFLECT_EVEN: s[d] = 2 * dhi - c + 1; v = A4(f, s[0], s[1], s[2], n); break;
            case ORC_BC_REFLECT_ODD: s[d] = 2 * dhi - c + 1; v = -A4(f, s[0], s[1], s[2], n); break;
            case ORC_BC_EXT_DIR: v = edhi; break;
            default: continue;
            }
        }
        A4(f, i, j, k, n) = v;
    }
}

void orc_fill_physbc_cc(orc_fab* f, const orc_geom* g, const orc_bcrec* bc,
                        const double* extdir_lo, const double* extdir_hi)
{
    for (int n = 0; n < f->nc; ++n)
        for (int d = 0; d < 3; ++d) {
            if (g->periodic[d]) continue;
            double el = extdir_lo ? extdir_lo[n * 3 + d] : 0.0;
            double eh = extdir_hi ? extdir_hi[n * 3 + d] : 0.0;
            physbc_dir(f, g, d, n, bc[n].lo[d], bc[n].hi[d], el, eh);
        }
}

/* ---- coarse-fine fill (amrex::FillPatchTwoLevels with the interpolater IAMR registers for State_Type / Gradp_Type, `cell_cons_interp`,
 * reference Source/NS_setup.cpp:206-394, = amrex::CellConservativeLinear constructed with do_linear_limiting = FALSE, i.e. the
 * per-component monotonised-central variant (upstream mf_cell_cons_lin_interp_mcslope).  Published algorithm restated, per component:
 * unlimited central slope dc (one-sided 4-point formula in the coarse cell next to an ext_dir / hoextrap domain face), limited
 * slope s_d = sign(dc) min(|dc|, 2|u(i+1)-u(i)|, 2|u(i)-u(i-1)|) or 0 at an extremum; then ONE factor alpha <= 1 for the three
 * slopes of the component so that the largest excursion inside the coarse cell, sum_d |s_d| (r-1)/(2r), stays within the minimum /
 * maximum of the 27 coarse neighbours; fine = crse + sum_d offset_d * alpha * s_d.
 * (Round 1 restated the linear-limiting variant `lincc_interp`, whose limiter factor is shared by all components; a component
 * that is zero up to round-off noise -- w in a two-dimensional flow -- then switches the slopes of every component off.)
 * crse: coarse level data with >= 1 filled ghost cell (periodic / physical BC already applied), covering what is needed.
 * Fills every cell of `fine` (incl. ghosts) that lies inside [flo,fhi] (the region to fill) but OUTSIDE [vlo,vhi] (the fine
 * level's own valid box). */
static double cf_cslope(const orc_fab* u, const int c[3], int n, int d, int domlo, int domhi, int bclo, int bchi)
{
    int m[3] = {c[0], c[1], c[2]}, p[3] = {c[0], c[1], c[2]};
    m[d] -= 1; p[d] += 1;
    double dc = 0.5 * (A4(u, p[0], p[1], p[2], n) - A4(u, m[0], m[1], m[2], n));
    const double um = A4(u, m[0], m[1], m[2], n), u0 = A4(u, c[0], c[1], c[2], n), up = A4(u, p[0], p[1], p[2], n);
    if (c[d] == domlo && (bclo == ORC_BC_EXT_DIR || bclo == ORC_BC_HOEXTRAP)) {
        int pp[3] = {c[0], c[1], c[2]}; pp[d] += 2;
        if (pp[d] <= u->hi[d]) dc = -16. / 15. * um + 0.5 * u0 + 2. / 3. * up - 0.1 * A4(u, pp[0], pp[1], pp[2], n);
        else dc = 0.25 * (up + 5. * u0 - 6. * um);
    }
    if (c[d] == domhi && (bchi == ORC_BC_EXT_DIR || bchi == ORC_BC_HOEXTRAP)) {
        int mm[3] = {c[0], c[1], c[2]}; mm[d] -= 2;
        if (mm[d] >= u->lo[d]) dc = 16. / 15. * up - 0.5 * u0 - 2. / 3. * um + 0.1 * A4(u, mm[0], mm[1], mm[2], n);
        else dc = -0.25 * (um + 5. * u0 - 6. * up);
    }
    return dc;
}

void orc_fill_coarse_fine(orc_fab* fine, const int flo[3], const int fhi[3], const int vlo[3], const int vhi[3],
                          const orc_fab* crse, const int cdomlo[3], const int cdomhi[3], const int periodic[3], int ratio,
                          const orc_bcrec* bc)
{
    const int nc = fine->nc;
    const double exc = (double)(ratio - 1) / (double)(2 * ratio);
    for (int k = flo[2]; k <= fhi[2]; ++k) for (int j = flo[1]; j <= fhi[1]; ++j) for (int i = flo[0]; i <= fhi[0]; ++i) {
        if (i >= vlo[0] && i <= vhi[0] && j >= vlo[1] && j <= vhi[1] && k >= vlo[2] && k <= vhi[2]) continue;
        const int f[3] = {i, j, k};
        int c[3];
        double off[3];
        for (int d = 0; d < 3; ++d) {
            c[d] = f[d] >= 0 ? f[d] / ratio : -((-f[d] + ratio - 1) / ratio);
            off[d] = ((double)(f[d] - c[d] * ratio) + 0.5) / (double)ratio - 0.5;
        }
        for (int n = 0; n < nc; ++n) {
            const double u0 = A4(crse, c[0], c[1], c[2], n);
            double sl[3];
            for (int d = 0; d < 3; ++d) {
                const int bl = periodic[d] ? ORC_BC_INT_DIR : bc[n].lo[d], bh = periodic[d] ? ORC_BC_INT_DIR : bc[n].hi[d];
                const double dc = cf_cslope(crse, c, n, d, cdomlo[d], cdomhi[d], bl, bh);
                int m[3] = {c[0], c[1], c[2]}, p[3] = {c[0], c[1], c[2]};
                m[d] -= 1; p[d] += 1;
                const double df = 2.0 * (A4(crse, p[0], p[1], p[2], n) - u0), db = 2.0 * (u0 - A4(crse, m[0], m[1], m[2], n));
                double s = (df * db >= 0.0) ? fmin(fabs(df), fabs(db)) : 0.0;
                sl[d] = copysign(1.0, dc) * fmin(s, fabs(dc));
            }
            double alpha = 1.0;
            if (sl[0] != 0.0 || sl[1] != 0.0 || sl[2] != 0.0) {
                const double dumax = fabs(sl[0]) * exc + fabs(sl[1]) * exc + fabs(sl[2]) * exc;
                double umax = u0, umin = u0;
                for (int ko = -1; ko <= 1; ++ko) for (int jo = -1; jo <= 1; ++jo) for (int io = -1; io <= 1; ++io) {
                    const double v = A4(crse, c[0] + io, c[1] + jo, c[2] + ko, n);
                    umin = fmin(umin, v); umax = fmax(umax, v);
                }
                if (dumax * alpha > (umax - u0)) alpha = (umax - u0) / dumax;
                if (dumax * alpha > (u0 - umin)) alpha = (u0 - umin) / dumax;
            }
            A4(fine, i, j, k, n) = u0 + off[0] * (sl[0] * alpha) + off[1] * (sl[1] * alpha) + off[2] * (sl[2] * alpha);
        }
    }
}

/* ---- NavierStokesBase::create_umac_grown on a refined level, one fine box (reference Source/NavierStokesBase.cpp:1108-1311):
 * ghost faces by FaceLinear interpolation of the coarse mac velocities (linear in the face-normal direction between the two
 * coarse faces, piecewise constant transversally), then the in-tree divergence fix: a ghost cell inside the domain with
 * exactly one face neighbour in the valid box gets its outer face reset so that div(u_mac) = 0 in it. */
void orc_create_umac_grown(orc_fab* uf[3], const int vlo[3], const int vhi[3], const orc_fab* uc[3], int ratio,
                           const double fdx[3], const int fdom_n[3], const int periodic[3])
{
    for (int d = 0; d < 3; ++d) {
        orc_fab* f = uf[d];
        for (int k = f->lo[2]; k <= f->hi[2]; ++k) for (int j = f->lo[1]; j <= f->hi[1]; ++j) for (int i = f->lo[0]; i <= f->hi[0]; ++i) {
            const int fi[3] = {i, j, k};
            int valid = 1;
            for (int e = 0; e < 3; ++e) if (fi[e] < vlo[e] || fi[e] > vhi[e] + (e == d ? 1 : 0)) valid = 0;
            if (valid) continue;
            int c[3];
            for (int e = 0; e < 3; ++e) c[e] = fi[e] >= 0 ? fi[e] / ratio : -((-fi[e] + ratio - 1) / ratio);
            const int rem = fi[d] - c[d] * ratio;
            double v;
            if (rem == 0) v = A4(uc[d], c[0], c[1], c[2], 0);
            else {
                const double w = (double)rem / (double)ratio;
                int cp[3] = {c[0], c[1], c[2]}; cp[d] += 1;
                v = (1.0 - w) * A4(uc[d], c[0], c[1], c[2], 0) + w * A4(uc[d], cp[0], cp[1], cp[2], 0);
            }
            A4(f, i, j, k, 0) = v;
        }
    }
    /* level mask on the box grown by 2: 0 interior, 2 not covered, 3 outside the physical domain */
    #define MASK(i, j, k) mask_of(i, j, k, vlo, vhi, fdom_n, periodic)
    for (int k = vlo[2] - 1; k <= vhi[2] + 1; ++k) for (int j = vlo[1] - 1; j <= vhi[1] + 1; ++j) for (int i = vlo[0] - 1; i <= vhi[0] + 1; ++i) {
        int idx[3] = {i, j, k};
        int m = 0, outside = 0;
        for (int e = 0; e < 3; ++e) { if (idx[e] < vlo[e] || idx[e] > vhi[e]) m = 2; if (!periodic[e] && (idx[e] < 0 || idx[e] > fdom_n[e] - 1)) outside = 1; }
        if (m != 2 || outside) continue;
        int count = 0;
        for (int e = 0; e < 3; ++e)
            for (int s = -1; s <= 1; s += 2) {
                int q[3] = {i, j, k}; q[e] += s;
                int in = 1;
                for (int r = 0; r < 3; ++r) if (q[r] < vlo[r] || q[r] > vhi[r]) in = 0;
                count += in;
            }
        if (count != 1) continue;
        orc_fab *u = uf[0], *v = uf[1], *w = uf[2];
        const double dux = (A4(u, i + 1, j, k, 0) - A4(u, i, j, k, 0)) / fdx[0];
        const double duy = (A4(v, i, j + 1, k, 0) - A4(v, i, j, k, 0)) / fdx[1];
        const double duz = (A4(w, i, j, k + 1, 0) - A4(w, i, j, k, 0)) / fdx[2];
        /* "m(i+1,j,k) != notcovered" for a single box == that neighbour is interior */
        #define INBOX(a, b, c) ((a) >= vlo[0] && (a) <= vhi[0] && (b) >= vlo[1] && (b) <= vhi[1] && (c) >= vlo[2] && (c) <= vhi[2])
        if (i < vlo[0] && INBOX(i + 1, j, k)) A4(u, i, j, k, 0) = A4(u, i + 1, j, k, 0) + fdx[0] * (duy + duz - 0.0);
        else if (i > vhi[0] && INBOX(i - 1, j, k)) A4(u, i + 1, j, k, 0) = A4(u, i, j, k, 0) - fdx[0] * (duy + duz - 0.0);
        if (j < vlo[1] && INBOX(i, j + 1, k)) A4(v, i, j, k, 0) = A4(v, i, j + 1, k, 0) + fdx[1] * (dux + duz - 0.0);
        else if (j > vhi[1] && INBOX(i, j - 1, k)) A4(v, i, j + 1, k, 0) = A4(v, i, j, k, 0) - fdx[1] * (dux + duz - 0.0);
        if (k < vlo[2] && INBOX(i, j, k + 1)) A4(w, i, j, k, 0) = A4(w, i, j, k + 1, 0) + fdx[2] * (dux + duy - 0.0);
        else if (k > vhi[2] && INBOX(i, j, k - 1)) A4(w, i, j, k + 1, 0) = A4(w, i, j, k, 0) - fdx[2] * (dux + duy - 0.0);
        #undef INBOX
    }
    #undef MASK
}
