/* oracle/orc_syncreg.c -- IAMR's SyncRegister restated LITERALLY, box by box (TEST INFRASTRUCTURE ONLY, see orc.h).
 *
 * This is the one piece of the multi-level path whose source is in the reference tree, so it is followed as written
 * (/root/reference/Source/SyncRegister.cpp, SyncRegister.H):
 *   constructor  SyncRegister.cpp:18-45    grids = coarsened fine boxes; bndry[face] / bndry_mask[face] = one nodal fab per box on
 *                                           the face plane of its nodal box (BndryBATransformer(face, NodeType, 0, 1, 0))
 *   InitRHS      SyncRegister.cpp:47-285   rhs = 0; copyTo of every register (periodic); zero on outflow faces; bndry_mask = 8-cell
 *                                           count of cells under the fine grids (periodic images), x 2 per non-periodic domain face the
 *                                           node lies on, -> 0 where the count exceeds maxcount = DIM^DIM - 0.5 (26.5 in 3-D: never,
 *                                           see orc_syncreg_init_rhs), else 1; rhs *= mask of every face
 *   CrseInit     SyncRegister.cpp:287-300  setVal(0); resid *= mult; bndry[face].plusFrom(resid) (periodic)
 *   CompAdd      SyncRegister.cpp:302-348  zero the fine residual under Pgrids (and their periodic images), then FineAdd
 *   FineAdd      SyncRegister.cpp:350-607  resid *= mult; edges of every fine nodal box x 1/2, corners x 2/3 (:369-425); per
 *                                           direction and side the coarse face plane of the box receives the in-plane weighted sum
 *                                           coeff = (r-m)(r-n) r_dir / (r0^2 r1^2 r2^2), x 1/2 for m == 0 and for n == 0, of the four
 *                                           fine nodes (+-m, +-n) (:447-525); x 2 per non-periodic domain face the coarse node lies
 *                                           on (:526-560); the six planes are summed into a nodal fab on the coarsened box (:563-573);
 *                                           the scaling of the fine fab is undone (:577-625); bndry[face].plusFrom(sum) (periodic)
 * amrex::FabSet::plusFrom / copyTo are FabArray::ParallelCopy with op ADD / COPY between two box arrays: every source fab (valid nodal
 * box, overlapping its neighbours on shared faces) is added to / copied into every destination fab it intersects, once per periodic shift.
 *
 * The residuals that feed the registers are upstream's (amrex::MLNodeLaplacian::compSyncResidualCoarse / compSyncResidualFine through
 * Hydro::NodalProjector, absent from the reference tree): they are formed BOX BY BOX from the cells of that box only -- which is what
 * makes the overlapping-box sums above come out right -- and are restated here from their published behaviour (parity unpinned):
 *   fine  : on the nodal box of every fine box, rhs - L(phi) with velocity and sigma of the box's own cells (zero outside), NO Neumann
 *           doubling at physical walls (SyncRegister::FineAdd doubles), zero on Dirichlet nodes
 *   coarse: on the nodal box of every coarse box, with the box's own cells that the fine level does not cover, only at nodes that touch
 *           both covered and uncovered cells of the LEVEL (periodic images, mirror across walls), doubled per Neumann wall, zero elsewhere
 * The single-valued "union" restatement in orc_amr.c (syncreg_*) is kept as the cross-check: tests/test_cpu_syncreg.py. */
#include "orc_ns_int.h"

typedef struct orc_ndmf {      /* nodal MultiFab: one fab per cell-centred box, on its nodal box grown by ng */
    int nbox;
    int* boxes;                /* 6 ints per box: lo[3], hi[3] of the cell box */
    int ng;
    orc_fab* fab;
} orc_ndmf;

struct orc_syncreg {
    int nbox;                  /* coarsened fine boxes */
    int* cboxes;
    int ratio;
    orc_fab* bndry[6];         /* [face = dir + 3 * side][box] */
    orc_fab* mask[6];
};

static orc_fab fab_on(const int lo[3], const int hi[3])
{
    orc_fab f;
    for (int d = 0; d < 3; ++d) { f.lo[d] = lo[d]; f.hi[d] = hi[d]; }
    f.nc = 1;
    f.p = (double*)calloc(orc_npts(&f), sizeof(double));
    return f;
}

orc_ndmf* orc_ndmf_create(int nbox, const int* boxes, int ng)
{
    orc_ndmf* m = (orc_ndmf*)calloc(1, sizeof(orc_ndmf));
    m->nbox = nbox; m->ng = ng;
    m->boxes = (int*)malloc(sizeof(int) * 6 * (size_t)nbox);
    memcpy(m->boxes, boxes, sizeof(int) * 6 * (size_t)nbox);
    m->fab = (orc_fab*)calloc((size_t)nbox, sizeof(orc_fab));
    for (int b = 0; b < nbox; ++b) {
        int lo[3], hi[3];
        for (int d = 0; d < 3; ++d) { lo[d] = boxes[6 * b + d] - ng; hi[d] = boxes[6 * b + 3 + d] + 1 + ng; }
        m->fab[b] = fab_on(lo, hi);
    }
    return m;
}
void orc_ndmf_destroy(orc_ndmf* m)
{
    if (!m) return;
    for (int b = 0; b < m->nbox; ++b) orc_free(&m->fab[b]);
    free(m->fab); free(m->boxes); free(m);
}
double* orc_ndmf_fab(orc_ndmf* m, int b, int lo[3], int hi[3])
{
    for (int d = 0; d < 3; ++d) { lo[d] = m->fab[b].lo[d]; hi[d] = m->fab[b].hi[d]; }
    return m->fab[b].p;
}
void orc_ndmf_setval(orc_ndmf* m, double v) { for (int b = 0; b < m->nbox; ++b) orc_setval(&m->fab[b], v); }

/* valid (nodal) box of fab b */
static void valid_box(const orc_ndmf* m, int b, int lo[3], int hi[3])
{
    for (int d = 0; d < 3; ++d) { lo[d] = m->boxes[6 * b + d]; hi[d] = m->boxes[6 * b + 3 + d] + 1; }
}

/* the periodic shifts of amrex::Periodicity::shiftIntVect (the zero shift included) */
static int periodic_shifts(const orc_geom* g, int sh[27][3])
{
    int n = 0;
    for (int sz = -1; sz <= 1; ++sz) for (int sy = -1; sy <= 1; ++sy) for (int sx = -1; sx <= 1; ++sx) {
        const int s[3] = {sx, sy, sz};
        int ok = 1;
        for (int d = 0; d < 3; ++d) if (s[d] != 0 && !(g && g->periodic[d])) ok = 0;
        if (!ok) continue;
        for (int d = 0; d < 3; ++d) sh[n][d] = s[d] * (g ? g->n[d] : 0);
        ++n;
    }
    return n;
}

/* FabArray::ParallelCopy(dst <- src valid boxes, ngrow 0, periodicity, op): dst(p) (+)= src(p + shift) wherever p lies in the dst fab
 * and p + shift in the valid box of a src fab; add: every hit adds, copy: the hits overwrite each other */
static void pcopy(orc_fab* dst, const orc_ndmf* src, const orc_geom* g, int add)
{
    int sh[27][3];
    const int ns = periodic_shifts(g, sh);
    for (int b = 0; b < src->nbox; ++b) {
        int slo[3], shi[3];
        valid_box(src, b, slo, shi);
        for (int q = 0; q < ns; ++q) {
            int lo[3], hi[3], ok = 1;
            for (int d = 0; d < 3; ++d) {      /* dst index p = src index - shift */
                lo[d] = slo[d] - sh[q][d] > dst->lo[d] ? slo[d] - sh[q][d] : dst->lo[d];
                hi[d] = shi[d] - sh[q][d] < dst->hi[d] ? shi[d] - sh[q][d] : dst->hi[d];
                if (hi[d] < lo[d]) ok = 0;
            }
            if (!ok) continue;
            for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) {
                const double v = A4(&src->fab[b], i + sh[q][0], j + sh[q][1], k + sh[q][2], 0);
                if (add) A4(dst, i, j, k, 0) += v; else A4(dst, i, j, k, 0) = v;
            }
        }
    }
}
/* the other direction: every register fab into the fabs of a MultiFab (FabSet::copyTo) */
static void pcopy_to(orc_ndmf* dst, const orc_fab* src, const orc_geom* g)
{
    int sh[27][3];
    const int ns = periodic_shifts(g, sh);
    for (int b = 0; b < dst->nbox; ++b) {
        orc_fab* D = &dst->fab[b];
        int dlo[3], dhi[3];
        valid_box(dst, b, dlo, dhi);
        for (int d = 0; d < 3; ++d) { dlo[d] -= dst->ng; dhi[d] += dst->ng; }       /* copyTo(rhs, ngrow, ...) */
        for (int q = 0; q < ns; ++q) {
            int lo[3], hi[3], ok = 1;
            for (int d = 0; d < 3; ++d) {
                lo[d] = src->lo[d] - sh[q][d] > dlo[d] ? src->lo[d] - sh[q][d] : dlo[d];
                hi[d] = src->hi[d] - sh[q][d] < dhi[d] ? src->hi[d] - sh[q][d] : dhi[d];
                if (hi[d] < lo[d]) ok = 0;
            }
            if (!ok) continue;
            for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i)
                A4(D, i, j, k, 0) = A4(src, i + sh[q][0], j + sh[q][1], k + sh[q][2], 0);
        }
    }
}

/* SyncRegister::SyncRegister (SyncRegister.cpp:18-45) */
orc_syncreg* orc_syncreg_create(int nbox, const int* fine_boxes, int ratio)
{
    orc_syncreg* sr = (orc_syncreg*)calloc(1, sizeof(orc_syncreg));
    sr->nbox = nbox; sr->ratio = ratio;
    sr->cboxes = (int*)malloc(sizeof(int) * 6 * (size_t)nbox);
    for (int b = 0; b < nbox; ++b) for (int d = 0; d < 3; ++d) {
        const int lo = fine_boxes[6 * b + d], hi = fine_boxes[6 * b + 3 + d];
        sr->cboxes[6 * b + d] = lo >= 0 ? lo / ratio : -((-lo + ratio - 1) / ratio);
        sr->cboxes[6 * b + 3 + d] = hi >= 0 ? hi / ratio : -((-hi + ratio - 1) / ratio);
    }
    for (int dir = 0; dir < 3; ++dir) for (int side = 0; side < 2; ++side) {
        const int f = dir + 3 * side;
        sr->bndry[f] = (orc_fab*)calloc((size_t)nbox, sizeof(orc_fab));
        sr->mask[f] = (orc_fab*)calloc((size_t)nbox, sizeof(orc_fab));
        for (int b = 0; b < nbox; ++b) {
            int lo[3], hi[3];
            for (int d = 0; d < 3; ++d) { lo[d] = sr->cboxes[6 * b + d]; hi[d] = sr->cboxes[6 * b + 3 + d] + 1; }   /* nodal box */
            if (side == 0) hi[dir] = lo[dir]; else lo[dir] = hi[dir];                                                   /* bdryNode plane */
            sr->bndry[f][b] = fab_on(lo, hi);
            sr->mask[f][b] = fab_on(lo, hi);
        }
    }
    return sr;
}
void orc_syncreg_destroy(orc_syncreg* sr)
{
    if (!sr) return;
    for (int f = 0; f < 6; ++f) {
        for (int b = 0; b < sr->nbox; ++b) { orc_free(&sr->bndry[f][b]); orc_free(&sr->mask[f][b]); }
        free(sr->bndry[f]); free(sr->mask[f]);
    }
    free(sr->cboxes); free(sr);
}
void orc_syncreg_setval(orc_syncreg* sr, double v)
{
    for (int f = 0; f < 6; ++f) for (int b = 0; b < sr->nbox; ++b) orc_setval(&sr->bndry[f][b], v);
}

/* SyncRegister::CrseInit (SyncRegister.cpp:287-300); resid: nodal MultiFab on the coarse level's boxes (scaled in place, as upstream) */
void orc_syncreg_crse_init(orc_syncreg* sr, orc_ndmf* resid_crse, const orc_geom* cgeom, double mult)
{
    orc_syncreg_setval(sr, 0.0);
    for (int b = 0; b < resid_crse->nbox; ++b) { const size_t N = orc_npts(&resid_crse->fab[b]); for (size_t q = 0; q < N; ++q) resid_crse->fab[b].p[q] *= mult; }
    for (int f = 0; f < 6; ++f) for (int b = 0; b < sr->nbox; ++b) pcopy(&sr->bndry[f][b], resid_crse, cgeom, 1);
}

/* the edge / corner scaling of a fine nodal box (SyncRegister.cpp:369-425 and its inverse :577-625) */
static void scale_edges_corners(orc_fab* F, const int flo[3], const int fhi[3], double edge, double corner)
{
    for (int k = flo[2]; k <= fhi[2]; ++k) for (int j = flo[1]; j <= fhi[1]; ++j) for (int i = flo[0]; i <= fhi[0]; ++i) {
        const int onx = i == flo[0] || i == fhi[0], ony = j == flo[1] || j == fhi[1], onz = k == flo[2] || k == fhi[2];
        if (onx + ony + onz >= 2) A4(F, i, j, k, 0) *= edge;          /* the twelve edges, end points included */
        if (onx + ony + onz == 3) A4(F, i, j, k, 0) *= corner;        /* the eight corners */
    }
}

/* SyncRegister::FineAdd (SyncRegister.cpp:350-607); resid: nodal MultiFab on the FINE boxes with ngrow >= ratio - 1 (its ghost nodes are
 * read as they are -- zero, Projection.cpp:375-376) */
void orc_syncreg_fine_add(orc_syncreg* sr, orc_ndmf* resid_fine, const orc_geom* cgeom, double mult)
{
    const int r = sr->ratio;
    for (int b = 0; b < resid_fine->nbox; ++b) { const size_t N = orc_npts(&resid_fine->fab[b]); for (size_t q = 0; q < N; ++q) resid_fine->fab[b].p[q] *= mult; }
    /* Sync_resid_crse on the coarsened (nodal) boxes of the fine MultiFab */
    int* cb = (int*)malloc(sizeof(int) * 6 * (size_t)resid_fine->nbox);
    for (int b = 0; b < resid_fine->nbox; ++b) for (int d = 0; d < 3; ++d) {
        cb[6 * b + d] = resid_fine->boxes[6 * b + d] / r;                              /* aligned, non-negative index spaces */
        cb[6 * b + 3 + d] = (resid_fine->boxes[6 * b + 3 + d] + 1) / r - 1;
    }
    orc_ndmf* crse = orc_ndmf_create(resid_fine->nbox, cb, 0);
    free(cb);
    const double twoThirds = 2.0 / 3.0, threeHalves = 3.0 / 2.0;
    for (int b = 0; b < resid_fine->nbox; ++b) {
        orc_fab* F = &resid_fine->fab[b];
        int flo[3], fhi[3];
        valid_box(resid_fine, b, flo, fhi);
        scale_edges_corners(F, flo, fhi, 0.5, twoThirds);
        orc_fab* C = &crse->fab[b];
        for (int dir = 0; dir < 3; ++dir) {
            const int dim1 = dir != 0 ? 0 : 1, dim2 = dir != 0 ? (dir == 2 ? 1 : 2) : 2;
            for (int side = 0; side < 2; ++side) {
                int lo[3], hi[3];
                for (int d = 0; d < 3; ++d) { lo[d] = C->lo[d]; hi[d] = C->hi[d]; }
                if (side == 0) hi[dir] = lo[dir]; else lo[dir] = hi[dir];
                const double denom = (double)r / (double)((long)r * r * r * r * r * r);
                for (int kc = lo[2]; kc <= hi[2]; ++kc) for (int jc = lo[1]; jc <= hi[1]; ++jc) for (int ic = lo[0]; ic <= hi[0]; ++ic) {
                    const int idxc[3] = {ic, jc, kc};
                    double v = 0.0;
                    for (int n = 0; n < r; ++n) for (int m = 0; m < r; ++m) {
                        double coeff = (double)(r - m) * (double)(r - n) * denom;
                        if (n == 0) coeff *= 0.5;
                        if (m == 0) coeff *= 0.5;
                        int f0[3], f1[3], f2[3], f3[3];
                        for (int d = 0; d < 3; ++d) f0[d] = f1[d] = f2[d] = f3[d] = r * idxc[d];
                        f0[dim1] += m; f0[dim2] += n;
                        f1[dim1] -= m; f1[dim2] += n;
                        f2[dim1] += m; f2[dim2] -= n;
                        f3[dim1] -= m; f3[dim2] -= n;
                        v += coeff * (A4(F, f0[0], f0[1], f0[2], 0) + A4(F, f1[0], f1[1], f1[2], 0) + A4(F, f2[0], f2[1], f2[2], 0) + A4(F, f3[0], f3[1], f3[2], 0));
                    }
                    /* "points on the physical bndry must be doubled for any boundary but outflow or periodic" (:526-560) */
                    for (int n = 0; n < 3; ++n) {
                        if (cgeom->periodic[n]) continue;
                        if (idxc[n] == 0) v *= 2.0;
                        if (idxc[n] == cgeom->n[n]) v *= 2.0;
                    }
                    A4(C, ic, jc, kc, 0) += v;
                }
            }
        }
        scale_edges_corners(F, flo, fhi, 2.0, threeHalves);
    }
    for (int f = 0; f < 6; ++f) for (int b = 0; b < sr->nbox; ++b) pcopy(&sr->bndry[f][b], crse, cgeom, 1);
    orc_ndmf_destroy(crse);
}

/* SyncRegister::CompAdd (SyncRegister.cpp:302-348): Pgrids = nodal boxes (given as the cell boxes they surround) */
void orc_syncreg_comp_add(orc_syncreg* sr, orc_ndmf* resid_fine, const orc_geom* fgeom, const orc_geom* cgeom, int nP, const int* Pboxes, double mult)
{
    int sh[27][3];
    const int ns = periodic_shifts(fgeom, sh);
    for (int b = 0; b < resid_fine->nbox; ++b) {
        int slo[3], shi[3];
        valid_box(resid_fine, b, slo, shi);
        for (int p = 0; p < nP; ++p) for (int q = 0; q < ns; ++q) {
            int lo[3], hi[3], ok = 1;
            for (int d = 0; d < 3; ++d) {
                const int plo = Pboxes[6 * p + d] + sh[q][d], phi = Pboxes[6 * p + 3 + d] + 1 + sh[q][d];
                lo[d] = plo > slo[d] ? plo : slo[d]; hi[d] = phi < shi[d] ? phi : shi[d];
                if (hi[d] < lo[d]) ok = 0;
            }
            if (!ok) continue;
            for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) A4(&resid_fine->fab[b], i, j, k, 0) = 0.0;
        }
    }
    orc_syncreg_fine_add(sr, resid_fine, cgeom, mult);
}

static double g_maxcount = 3.0 * 3.0 * 3.0 - 0.5;
void orc_syncreg_set_maxcount(double m) { g_maxcount = m; }

/* SyncRegister::InitRHS (SyncRegister.cpp:47-285); rhs: nodal MultiFab on the coarse level's boxes */
void orc_syncreg_init_rhs(orc_syncreg* sr, orc_ndmf* rhs, const orc_geom* geom, const int phys_lo[3], const int phys_hi[3])
{
    orc_ndmf_setval(rhs, 0.0);
    for (int f = 0; f < 6; ++f) for (int b = 0; b < sr->nbox; ++b) pcopy_to(rhs, &sr->bndry[f][b], geom);
    /* outflow faces (:64-126) */
    for (int dir = 0; dir < 3; ++dir) for (int b = 0; b < rhs->nbox; ++b) {
        orc_fab* R = &rhs->fab[b];
        int lo[3], hi[3];
        valid_box(rhs, b, lo, hi);
        for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) {
            const int idx[3] = {i, j, k};
            if (phys_lo[dir] == 2 && idx[dir] == 0) A4(R, i, j, k, 0) = 0.0;              /* PhysBCType::outflow */
            if (phys_hi[dir] == 2 && idx[dir] == geom->n[dir]) A4(R, i, j, k, 0) = 0.0;
        }
    }
    /* bndry_mask (:128-262) */
    int sh[27][3];
    const int ns = periodic_shifts(geom, sh);
    for (int f = 0; f < 6; ++f) for (int b = 0; b < sr->nbox; ++b) {
        orc_fab* M = &sr->mask[f][b];
        int clo[3], chi[3];                                   /* mask_cells = enclosedCells(grow(fab.box(), 1)) */
        for (int d = 0; d < 3; ++d) { clo[d] = M->lo[d] - 1; chi[d] = M->hi[d]; }
        orc_fab tmp = fab_on(clo, chi);
        for (int g = 0; g < sr->nbox; ++g) for (int q = 0; q < ns; ++q) {
            /* zero shift: grids.intersections(mask_cells); other shifts only "if (!geom.Domain().contains(mask_cells))", which for
             * boxes inside the domain is the only case in which a shifted grid can reach mask_cells at all */
            int lo[3], hi[3], ok = 1;
            for (int d = 0; d < 3; ++d) {
                const int glo = sr->cboxes[6 * g + d] - sh[q][d], ghi = sr->cboxes[6 * g + 3 + d] - sh[q][d];
                lo[d] = glo > clo[d] ? glo : clo[d]; hi[d] = ghi < chi[d] ? ghi : chi[d];
                if (hi[d] < lo[d]) ok = 0;
            }
            if (!ok) continue;
            for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) A4(&tmp, i, j, k, 0) = 1.0;
        }
        for (int k = M->lo[2]; k <= M->hi[2]; ++k) for (int j = M->lo[1]; j <= M->hi[1]; ++j) for (int i = M->lo[0]; i <= M->hi[0]; ++i) {
            double s = A4(&tmp, i, j, k, 0) + A4(&tmp, i - 1, j, k, 0) + A4(&tmp, i, j - 1, k, 0) + A4(&tmp, i - 1, j - 1, k, 0)
                     + A4(&tmp, i, j, k - 1, 0) + A4(&tmp, i - 1, j, k - 1, 0) + A4(&tmp, i, j - 1, k - 1, 0) + A4(&tmp, i - 1, j - 1, k - 1, 0);
            const int idx[3] = {i, j, k};
            for (int d = 0; d < 3; ++d) {                     /* "double the cell contributions if at a non-periodic physical bdry" */
                if (geom->periodic[d]) continue;
                if (idx[d] == 0) s *= 2.0;
                if (idx[d] == geom->n[d]) s *= 2.0;
            }
            /* "convert from sum of cell contributions to 0 or 1" (:264-283): maxcount = AMREX_D_TERM(DIM, *DIM, *DIM) - 0.5, i.e.
             * 3*3*3 - 0.5 = 26.5 in three dimensions (2*2 - 0.5 = 3.5 in two, where DIM^DIM happens to equal the 2^DIM cells around
             * a node).  Followed as written: in 3-D a count of at most 8 (x 2 per wall with the cells outside the domain uncounted:
             * still at most 8) never exceeds 26.5, so NO node is masked.  The nodes the 2-D form masks -- those surrounded by fine
             * cells only -- carry the sum of the per-box residual pieces of an interior node of the fine level (the converged level
             * residual, i.e. solver-tolerance noise) and lie under the fine level, where the composite solve of MLsyncProject ignores
             * the coarse right-hand side; orc_syncreg_set_maxcount(7.5) gives the 2-D behaviour for the cross-check with the
             * single-valued restatement (orc_amr.c), which zeroes them. */
            A4(M, i, j, k, 0) = s > g_maxcount ? 0.0 : 1.0;
        }
        orc_free(&tmp);
    }
    /* rhs *= mask of every face; the masks are copied WITHOUT periodicity (:276-284) */
    for (int f = 0; f < 6; ++f) {
        orc_ndmf* tmp = orc_ndmf_create(rhs->nbox, rhs->boxes, rhs->ng);
        orc_ndmf_setval(tmp, 1.0);
        for (int b = 0; b < sr->nbox; ++b) pcopy_to(tmp, &sr->mask[f][b], NULL);
        for (int b = 0; b < rhs->nbox; ++b) { const size_t N = orc_npts(&rhs->fab[b]); for (size_t q = 0; q < N; ++q) rhs->fab[b].p[q] *= tmp->fab[b].p[q]; }
        orc_ndmf_destroy(tmp);
    }
}

/* read access for the tests: register fab of (face = dir + 3 * side, box) */
double* orc_syncreg_fab(orc_syncreg* sr, int face, int b, int lo[3], int hi[3])
{
    for (int d = 0; d < 3; ++d) { lo[d] = sr->bndry[face][b].lo[d]; hi[d] = sr->bndry[face][b].hi[d]; }
    return sr->bndry[face][b].p;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * The box-by-box sync residuals (upstream amrex::MLNodeLaplacian::compSyncResidualFine / compSyncResidualCoarse, restated from their
 * published behaviour -- see the header).  Level data are the whole-domain arrays of orc_ns_int.h; the element-by-element form of
 * the operator is that of orc_nodal.c. */
#define NEUW(b) ((b) == ORC_LO_NEUMANN || (b) == ORC_LO_INFLOW)
static inline double elem_w(int a, int b, const double* dx)
{
    int ax = a & 1, ay = (a >> 1) & 1, az = (a >> 2) & 1;
    int bx = b & 1, by = (b >> 1) & 1, bz = (b >> 2) & 1;
    double sx = ax == bx ? 1. : -1., sy = ay == by ? 1. : -1., sz = az == bz ? 1. : -1.;
    double mx = ax == bx ? 1. / 3. : 1. / 6., my = ay == by ? 1. / 3. : 1. / 6., mz = az == bz ? 1. / 3. : 1. / 6.;
    return -(sx / (dx[0] * dx[0]) * my * mz + sy / (dx[1] * dx[1]) * mx * mz + sz / (dx[2] * dx[2]) * mx * my);
}
static int in_box(const int* bx, int i, int j, int k) { return i >= bx[0] && i <= bx[3] && j >= bx[1] && j <= bx[4] && k >= bx[2] && k <= bx[5]; }
static int level_nbox(const orc_ns_state* s) { return s->level == 0 ? 1 : s->nbox; }
static void level_box(const orc_ns_state* s, int b, int bx[6])
{
    if (s->level == 0) { for (int d = 0; d < 3; ++d) { bx[d] = 0; bx[3 + d] = s->g.n[d] - 1; } }
    else memcpy(bx, s->boxes + 6 * b, sizeof(int) * 6);
}
static int on_dirichlet(const orc_ns_state* s, int i, int j, int k)
{
    const int idx[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) {
        if (s->g.periodic[d]) continue;
        if (idx[d] == 0 && s->nlobc[d] == ORC_LO_DIRICHLET) return 1;
        if (idx[d] == s->g.n[d] && s->nhibc[d] == ORC_LO_DIRICHLET) return 1;
    }
    return 0;
}
/* what the cells of box bx (own cells only; `skip`: cell predicate of cells that do not count, may be NULL) contribute to
 * rhs - L(phi) at node (i,j,k): div(vold) of mlndlap_divu minus the element rows of the operator; cells beyond an inflow face next to
 * an own cell contribute their normal velocity (set_boundary_velocity, Source/Projection.cpp:2570-2663) */
static double box_resid_at(const orc_ns_state* s, const int* bx, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig,
                           int (*skip)(const orc_ns_state*, int, int, int), int i, int j, int k, const orc_fab* rhcc)
{
    const orc_geom* g = &s->g;
    double r = 0.0;
    for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
        const int c[3] = {cx, cy, cz};
        const int cell[3] = {i - 1 + cx, j - 1 + cy, k - 1 + cz};
        int nout = 0, eout = -1;
        for (int e = 0; e < 3; ++e) if (!g->periodic[e] && (cell[e] < 0 || cell[e] > g->n[e] - 1)) { ++nout; eout = e; }
        if (nout == 1) {
            const int bt = cell[eout] < 0 ? s->nlobc[eout] : s->nhibc[eout];
            if (bt != ORC_LO_INFLOW) continue;
            int in[3] = {cell[0], cell[1], cell[2]};
            in[eout] += cell[eout] < 0 ? 1 : -1;
            if (!in_box(bx, in[0], in[1], in[2]) || (skip && skip(s, in[0], in[1], in[2]))) continue;
            r += 0.25 / g->dx[eout] * (c[eout] ? 1.0 : -1.0) * A4(vold, cell[0], cell[1], cell[2], eout);
            continue;
        }
        if (nout > 1 || !in_box(bx, cell[0], cell[1], cell[2]) || (skip && skip(s, cell[0], cell[1], cell[2]))) continue;
        for (int d = 0; d < 3; ++d) r += 0.25 / g->dx[d] * (c[d] ? 1.0 : -1.0) * A4(vold, cell[0], cell[1], cell[2], d);
        if (rhcc) r += 0.125 * A4(rhcc, cell[0], cell[1], cell[2], 0);            /* mlndlap_rhcc: the cell-centred source of the own cells */
        const double sg = A4(sig, cell[0], cell[1], cell[2], 0);
        const int a = (1 - cx) | ((1 - cy) << 1) | ((1 - cz) << 2);
        for (int b = 0; b < 8; ++b) r -= sg * elem_w(a, b, g->dx) * A4(phi, cell[0] + (b & 1), cell[1] + ((b >> 1) & 1), cell[2] + ((b >> 2) & 1), 0);
    }
    return r;
}

/* compSyncResidualFine of level s (> 0): nodal MultiFab on the level's boxes, ngrow = ratio - 1 (ghost nodes zero) */
orc_ndmf* orc_sync_resid_fine_boxes(const orc_ns_state* s, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig, const orc_fab* rhcc)
{
    const int nb = level_nbox(s);
    int* bxs = (int*)malloc(sizeof(int) * 6 * (size_t)nb);
    for (int b = 0; b < nb; ++b) level_box(s, b, bxs + 6 * b);
    orc_ndmf* m = orc_ndmf_create(nb, bxs, (s->ratio > 1 ? s->ratio : 2) - 1);
    for (int b = 0; b < nb; ++b) {
        const int* bx = bxs + 6 * b;
        for (int k = bx[2]; k <= bx[5] + 1; ++k) for (int j = bx[1]; j <= bx[4] + 1; ++j) for (int i = bx[0]; i <= bx[3] + 1; ++i)
            A4(&m->fab[b], i, j, k, 0) = on_dirichlet(s, i, j, k) ? 0.0 : box_resid_at(s, bx, vold, phi, sig, NULL, i, j, k, rhcc);
    }
    free(bxs);
    return m;
}

static int covered_by_fine(const orc_ns_state* s, int i, int j, int k)     /* own-cell predicate: the next finer level covers the cell */
{
    const int r = s->fine->ratio;
    return ns_covered(s->fine, r * i, r * j, r * k);
}
/* crse_cc_mask with one ghost cell: periodic images (ns_covered wraps) and mlndlap_fillbc_cc's mirror across non-periodic faces */
static int covered_mirror(const orc_ns_state* s, int i, int j, int k)
{
    int q[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) {
        if (s->g.periodic[d]) continue;
        if (q[d] < 0) q[d] = -q[d] - 1;
        else if (q[d] > s->g.n[d] - 1) q[d] = 2 * s->g.n[d] - 1 - q[d];
    }
    return covered_by_fine(s, q[0], q[1], q[2]);
}
/* compSyncResidualCoarse of level s (which has a finer level): nodal MultiFab on the level's boxes, one ghost node (Projection.cpp:368) */
orc_ndmf* orc_sync_resid_crse_boxes(const orc_ns_state* s, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig, const orc_fab* rhcc)
{
    const int nb = level_nbox(s);
    int* bxs = (int*)malloc(sizeof(int) * 6 * (size_t)nb);
    for (int b = 0; b < nb; ++b) level_box(s, b, bxs + 6 * b);
    orc_ndmf* m = orc_ndmf_create(nb, bxs, 1);
    for (int b = 0; b < nb; ++b) {
        const int* bx = bxs + 6 * b;
        for (int k = bx[2]; k <= bx[5] + 1; ++k) for (int j = bx[1]; j <= bx[4] + 1; ++j) for (int i = bx[0]; i <= bx[3] + 1; ++i) {
            int ncov = 0;
            for (int c = 0; c < 8; ++c) ncov += covered_mirror(s, i - 1 + (c & 1), j - 1 + ((c >> 1) & 1), k - 1 + ((c >> 2) & 1));
            double r = 0.0;
            if (ncov != 0 && ncov != 8 && !on_dirichlet(s, i, j, k)) {          /* mlndlap_crse_resid */
                r = box_resid_at(s, bx, vold, phi, sig, covered_by_fine, i, j, k, rhcc);
                const int idx[3] = {i, j, k};
                for (int d = 0; d < 3; ++d) {
                    if (s->g.periodic[d]) continue;
                    if (idx[d] == 0 && NEUW(s->nlobc[d])) r *= 2.0;
                    if (idx[d] == s->g.n[d] && NEUW(s->nhibc[d])) r *= 2.0;
                }
            }
            A4(&m->fab[b], i, j, k, 0) = r;
        }
    }
    free(bxs);
    return m;
}

/* the level's box list as a nodal MultiFab shell (for InitRHS) and the gather of such a MultiFab into a whole-domain nodal array */
orc_ndmf* orc_level_ndmf(const orc_ns_state* s, int ng)
{
    const int nb = level_nbox(s);
    int* bxs = (int*)malloc(sizeof(int) * 6 * (size_t)nb);
    for (int b = 0; b < nb; ++b) level_box(s, b, bxs + 6 * b);
    orc_ndmf* m = orc_ndmf_create(nb, bxs, ng);
    free(bxs);
    return m;
}
void orc_ndmf_to_domain(const orc_ndmf* m, orc_fab* out /* whole-domain nodal array */)
{
    for (int b = 0; b < m->nbox; ++b) {
        int lo[3], hi[3];
        valid_box(m, b, lo, hi);
        for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) for (int i = lo[0]; i <= hi[0]; ++i) A4(out, i, j, k, 0) = A4(&m->fab[b], i, j, k, 0);
    }
}
