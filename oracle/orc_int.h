/* oracle/orc_int.h -- internal helpers for the CPU oracle (test infrastructure only). */
#ifndef ORC_INT_H
#define ORC_INT_H
#include "orc.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>

static inline size_t orc_nx(const orc_fab* f) { return (size_t)(f->hi[0] - f->lo[0] + 1); }
static inline size_t orc_ny(const orc_fab* f) { return (size_t)(f->hi[1] - f->lo[1] + 1); }
static inline size_t orc_nz(const orc_fab* f) { return (size_t)(f->hi[2] - f->lo[2] + 1); }
static inline size_t orc_npts(const orc_fab* f) { return orc_nx(f) * orc_ny(f) * orc_nz(f); }

static inline size_t orc_off(const orc_fab* f, int i, int j, int k, int n)
{
    return (size_t)(i - f->lo[0]) + orc_nx(f) * ((size_t)(j - f->lo[1]) + orc_ny(f) * ((size_t)(k - f->lo[2]) + orc_nz(f) * (size_t)n));
}
#define A4(f, i, j, k, n) ((f)->p[orc_off((f), (i), (j), (k), (n))])

/* allocate a fab on cells [0,n-1] converted to `type` and grown by ng */
static inline orc_fab orc_alloc(const int n[3], const int type[3], int ng, int nc)
{
    orc_fab f;
    for (int d = 0; d < 3; ++d) { f.lo[d] = -ng; f.hi[d] = n[d] - 1 + (type ? type[d] : 0) + ng; }
    f.nc = nc;
    f.p = (double*)calloc(orc_npts(&f) * (size_t)nc, sizeof(double));
    return f;
}
static inline void orc_free(orc_fab* f) { free(f->p); f->p = NULL; }
static inline void orc_setval(orc_fab* f, double v)
{
    size_t N = orc_npts(f) * (size_t)f->nc;
    for (size_t q = 0; q < N; ++q) f->p[q] = v;
}
static inline void orc_copy_all(orc_fab* d, const orc_fab* s)
{
    memcpy(d->p, s->p, orc_npts(s) * (size_t)s->nc * sizeof(double));
}

static const int ORC_CELL[3] = {0, 0, 0};
static const int ORC_NODE[3] = {1, 1, 1};
static const int ORC_FACE[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};

#endif
