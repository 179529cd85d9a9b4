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
