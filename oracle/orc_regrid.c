/* oracle/orc_regrid.c -- CPU restatement of the error-estimation kernels of SURVEY row f1 (test infrastructure only; PARITY
 * UNPINNED, see orc.h): the "mag_vort" derived quantity (dermgvort, Source/NS_derive.cpp:86-264, non-EB branch) and the AMRErrorTag
 * tests that NavierStokes::error_setup / errorEst apply (Source/NS_error.cpp:10-145; upstream amrex::AMRErrorTag::operator()). */
#include "orc_int.h"

void orc_mag_vort(const orc_geom* g, orc_fab* out, const orc_fab* vel /* 3 comps, 1 ghost filled */)
{
    const double idx = 1.0 / g->dx[0], idy = 1.0 / g->dx[1], idz = 1.0 / g->dx[2];
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        const double vx = 0.5 * (A4(vel, i + 1, j, k, 1) - A4(vel, i - 1, j, k, 1)) * idx;
        const double wx = 0.5 * (A4(vel, i + 1, j, k, 2) - A4(vel, i - 1, j, k, 2)) * idx;
        const double uy = 0.5 * (A4(vel, i, j + 1, k, 0) - A4(vel, i, j - 1, k, 0)) * idy;
        const double wy = 0.5 * (A4(vel, i, j + 1, k, 2) - A4(vel, i, j - 1, k, 2)) * idy;
        const double uz = 0.5 * (A4(vel, i, j, k + 1, 0) - A4(vel, i, j, k - 1, 0)) * idz;
        const double vz = 0.5 * (A4(vel, i, j, k + 1, 1) - A4(vel, i, j, k - 1, 1)) * idz;
        A4(out, i, j, k, 0) = sqrt((wy - vz) * (wy - vz) + (uz - wx) * (uz - wx) + (vx - uy) * (vx - uy));
    }
}

/* mode 0 GREATER, 1 LESS, 2 VORT (value * 2^level), 3 GRAD; rb_lo/rb_hi optional RealBox; tagged cells := 1 */
void orc_error_tag(const orc_geom* g, orc_fab* tags, const orc_fab* f, int comp, int mode, double value, int level,
                   const double* rb_lo, const double* rb_hi)
{
    const double thr = mode == 2 ? value * pow(2.0, level) : value;
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        if (rb_lo && rb_hi) {
            const double x[3] = {g->problo[0] + (i + 0.5) * g->dx[0], g->problo[1] + (j + 0.5) * g->dx[1], g->problo[2] + (k + 0.5) * g->dx[2]};
            int in = 1;
            for (int d = 0; d < 3; ++d) if (x[d] < rb_lo[d] || x[d] > rb_hi[d]) in = 0;
            if (!in) continue;
        }
        const double v = A4(f, i, j, k, comp);
        int t;
        if (mode == 0 || mode == 2) t = v >= thr;
        else if (mode == 1) t = v <= thr;
        else {
            double m = fabs(A4(f, i + 1, j, k, comp) - v);
            m = fmax(m, fabs(v - A4(f, i - 1, j, k, comp)));
            m = fmax(m, fabs(A4(f, i, j + 1, k, comp) - v));
            m = fmax(m, fabs(v - A4(f, i, j - 1, k, comp)));
            m = fmax(m, fabs(A4(f, i, j, k + 1, comp) - v));
            m = fmax(m, fabs(v - A4(f, i, j, k - 1, comp)));
            t = m >= thr;
        }
        if (t) A4(tags, i, j, k, 0) = 1.0;
    }
}
