/* oracle/orc_amr.c -- the multi-level time step of IAMR restated on the CPU: subcycled advance of a hierarchy of levels, flux
 * registers, reflux, average down, MAC sync, sync registers, the multi-level (composite) nodal projections and the multi-level
 * initialisation sequence.  TEST INFRASTRUCTURE ONLY (see orc.h); PARITY UNPINNED for the upstream parts.
 *
 * In-tree orchestration followed line by line (all in /root/reference/Source):
 *   Amr::timeStep / coarseTimeStep (upstream AMReX: advance, ncycle x timeStep(level+1), post_timestep)
 *   NavierStokesBase::post_timestep        NavierStokesBase.cpp:2546-2636
 *   NavierStokes::reflux                   NavierStokes.cpp:1736-1838
 *   NavierStokes::avgDown / avgDown_StatePress NavierStokes.cpp:1845-1873, NavierStokesBase.cpp:4125-4163
 *   NavierStokes::mac_sync                 NavierStokes.cpp:1438-1730
 *   MacProj::mac_sync_solve / mac_sync_compute MacProj.cpp:359-479, 490-731
 *   NavierStokesBase::level_sync           NavierStokesBase.cpp:1927-2044
 *   Projection::MLsyncProject              Projection.cpp:457-607
 *   SyncRegister::{InitRHS,CrseInit,FineAdd} SyncRegister.cpp:47-607
 *   NavierStokesBase::SyncInterp / SyncProjInterp NavierStokesBase.cpp:3071-3341
 *   Projection::initialVelocityProject / initialPressureProject / initialSyncProject Projection.cpp:615-1185
 *   NavierStokes::post_init / post_init_press, NSB::post_init_state / post_init_estDT NavierStokes.cpp:1254-1432, NavierStokesBase.cpp:2307-2439
 *   NavierStokesBase::computeNewDt         NavierStokesBase.cpp:945-1035
 * Upstream pieces restated from their published behaviour (AMReX FluxRegister / YAFluxRegister, MLNodeLaplacian multi-level
 * composite operator with its sync residuals, Hydro::NodalProjector): see the comments at each function.
 *
 * The composite nodal system is solved here by a conjugate-gradient iteration on the conforming finite-element composite
 * operator (hanging nodes on a coarse/fine boundary are slaves of the coarse nodes) -- deliberately NOT the multigrid cycle of the
 * product, so that agreement of the two is a check of the composite discretisation, not of a shared solver.
 * Scope: ratio 2, inviscid or viscous velocity on level 0 only where stated, non-diffusive scalars on refined hierarchies. */
#include "orc_ns_int.h"

struct orc_amr {
    int nlev;
    orc_ns_state* lev[8];
    int n_cycle[8];
    double dt_level[8], dt_min[8];
    int level_steps;
    double stop_time;
    orc_mg_stats st_sync;         /* last MLsyncProject */
    int sync_iters;
};
typedef struct orc_amr orc_amr;

void amr_composite_project_rhcc(orc_amr* a, int c0, int nl, orc_fab* vel[], orc_fab* phi[], const orc_fab* sig[], const orc_fab* rhnd,
                                orc_fab* const rhcc[], double rtol, double atol, int increment_gp, double inflow_scale, orc_mg_stats* st);
int orc_syncreg_literal = 1;
double orc_syncreg_diff_max = 0.0;
void orc_set_syncreg_literal(int on) { orc_syncreg_literal = on; }
double orc_syncreg_last_diff(int reset) { const double v = orc_syncreg_diff_max; if (reset) orc_syncreg_diff_max = 0.0; return v; }

static inline int wrapi(int i, int n) { int r = i % n; return r < 0 ? r + n : r; }

/* is the cell (ci,cj,ck) of level f->crse covered by level f */
static int fine_covers(const orc_ns_state* f, int ci, int cj, int ck)
{
    const int r = f->ratio;
    return ns_covered(f, r * ci, r * cj, r * ck);
}
static int cell_in_domain(const orc_geom* g, int i, int j, int k)
{
    const int c[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) if (!g->periodic[d] && (c[d] < 0 || c[d] > g->n[d] - 1)) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Flux registers (amrex::FluxRegister; amrex::YAFluxRegister used as CrseInit(-dt F) / FineAdd(+dt F) / Reflux(scale 1)).
 * One value per coarse face on the coarse/fine boundary:
 *   CrseInit : reg = (+=) mult * coarse flux
 *   FineAdd  : reg += mult * sum of the ratio^2 fine fluxes of the coarse face
 *   Reflux   : the coarse cell OUTSIDE the fine level gets -scale*reg/vol if the face is its high face, +scale*reg/vol if it is
 *              its low face ("Reflux subtracts values at hi edge of coarse cell and adds values at lo edge", MacProj.cpp:398-400)
 * Periodic directions: the face at index n is the face at index 0 and is not visited. */
int reg_side(const orc_ns_state* f, int d, int i, int j, int k)
{
    const orc_geom* cg = &f->crse->g;
    const int idx[3] = {i, j, k};
    if (!cg->periodic[d] && (idx[d] <= 0 || idx[d] >= cg->n[d])) return 0;      /* a physical boundary is not a coarse/fine face */
    int m[3] = {i, j, k}; m[d] -= 1;
    const int lo = fine_covers(f, m[0], m[1], m[2]), hi = fine_covers(f, i, j, k);
    if (lo && !hi) return 1;
    if (!lo && hi) return -1;
    return 0;
}
void reg_setval(orc_fab reg[3], double v) { for (int d = 0; d < 3; ++d) orc_setval(&reg[d], v); }

#define FACE_LOOP(cg, d, i, j, k) \
    for (int k = 0; k <= (cg)->n[2] - ((d) == 2 ? ((cg)->periodic[2] ? 1 : 0) : 1); ++k) \
    for (int j = 0; j <= (cg)->n[1] - ((d) == 1 ? ((cg)->periodic[1] ? 1 : 0) : 1); ++j) \
    for (int i = 0; i <= (cg)->n[0] - ((d) == 0 ? ((cg)->periodic[0] ? 1 : 0) : 1); ++i)

void reg_crse_init(const orc_ns_state* f, orc_fab reg[3], const orc_fab* flux, int d, int sc, int dc, int nc, double mult, int add)
{
    const orc_geom* cg = &f->crse->g;
    FACE_LOOP(cg, d, i, j, k) {
        if (!reg_side(f, d, i, j, k)) continue;
        for (int n = 0; n < nc; ++n) {
            const double v = mult * A4(flux, i, j, k, sc + n);
            if (add) A4(&reg[d], i, j, k, dc + n) += v; else A4(&reg[d], i, j, k, dc + n) = v;
        }
    }
}
void reg_fine_add(const orc_ns_state* f, orc_fab reg[3], const orc_fab* flux, int d, int sc, int dc, int nc, double mult)
{
    const orc_geom* cg = &f->crse->g;
    const int r = f->ratio, d1 = (d + 1) % 3, d2 = (d + 2) % 3;
    FACE_LOOP(cg, d, i, j, k) {
        if (!reg_side(f, d, i, j, k)) continue;
        const int q[3] = {i, j, k};
        for (int n = 0; n < nc; ++n) {
            double s = 0.0;
            for (int b = 0; b < r; ++b) for (int a = 0; a < r; ++a) {
                int p[3];
                p[d] = r * q[d]; p[d1] = r * q[d1] + a; p[d2] = r * q[d2] + b;
                s += A4(flux, p[0], p[1], p[2], sc + n);
            }
            A4(&reg[d], i, j, k, dc + n) += mult * s;
        }
    }
}
void reg_reflux(const orc_ns_state* f, orc_fab reg[3], orc_fab* S, double volume, double scale, int sc, int dc, int nc)
{
    const orc_geom* cg = &f->crse->g;
    for (int d = 0; d < 3; ++d)
    FACE_LOOP(cg, d, i, j, k) {
        const int sg = reg_side(f, d, i, j, k);
        if (!sg) continue;
        int c[3] = {i, j, k};
        if (sg < 0) c[d] -= 1;
        for (int e = 0; e < 3; ++e) if (cg->periodic[e]) c[e] = wrapi(c[e], cg->n[e]);
        for (int n = 0; n < nc; ++n) A4(S, c[0], c[1], c[2], dc + n) += (double)sg * scale * A4(&reg[d], i, j, k, sc + n) / volume;
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Node classification of a level (composite operator, sync registers).  The cells around node (i,j,k) that lie outside a
 * non-periodic domain face do not count (the operator mirrors across Neumann walls). */
enum { ND_NONE = 0, ND_INTERIOR = 1, ND_BOUNDARY = 2 };
static int node_class(const orc_ns_state* s, int i, int j, int k)
{
    int nin = 0, ntot = 0;
    for (int c = 0; c < 8; ++c) {
        const int ci = i - 1 + (c & 1), cj = j - 1 + ((c >> 1) & 1), ck = k - 1 + ((c >> 2) & 1);
        if (!cell_in_domain(&s->g, ci, cj, ck)) continue;
        ++ntot; nin += ns_covered(s, ci, cj, ck);
    }
    if (nin == 0) return ND_NONE;
    return nin == ntot ? ND_INTERIOR : ND_BOUNDARY;
}
/* relation of node (i,j,k) of level c = f->crse to the finer level f: 0 untouched, 1 strictly inside, 2 on its boundary */
static int node_vs_fine(const orc_ns_state* f, int i, int j, int k)
{
    const orc_geom* cg = &f->crse->g;
    int nin = 0, ntot = 0;
    for (int c = 0; c < 8; ++c) {
        const int ci = i - 1 + (c & 1), cj = j - 1 + ((c >> 1) & 1), ck = k - 1 + ((c >> 2) & 1);
        if (!cell_in_domain(cg, ci, cj, ck)) continue;
        ++ntot; nin += fine_covers(f, ci, cj, ck);
    }
    if (nin == 0) return 0;
    return nin == ntot ? 1 : 2;
}
static int on_dirichlet_face(const orc_ns_state* s, int i, int j, int k)
{
    const int idx[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) {
        if (s->g.periodic[d]) continue;
        if (idx[d] == 0 && s->nlobc[d] == ORC_LO_DIRICHLET) return 1;
        if (idx[d] == s->g.n[d] && s->nhibc[d] == ORC_LO_DIRICHLET) return 1;
    }
    return 0;
}
/* node weight in sums: 0 for the periodic duplicate at index n, 1/2 per Neumann wall (MLNodeLinOp dot mask) */
static double node_wt(const orc_ns_state* s, int i, int j, int k)
{
    const int idx[3] = {i, j, k};
    double w = 1.0;
    for (int d = 0; d < 3; ++d) {
        if (s->g.periodic[d]) { if (idx[d] == s->g.n[d]) return 0.0; }
        else {
            if (idx[d] == 0 && s->nlobc[d] != ORC_LO_DIRICHLET) w *= 0.5;
            if (idx[d] == s->g.n[d] && s->nhibc[d] != ORC_LO_DIRICHLET) w *= 0.5;
        }
    }
    return w;
}

/* sigma of level s restricted to the cells of the level that the next finer level (if any, and if excl_fine) does not cover,
 * ghost cells filled (periodic images, mirror across walls) */
static orc_fab masked_sigma(const orc_ns_state* s, const orc_fab* sig, int excl_fine)
{
    const orc_geom* g = &s->g;
    orc_fab m = orc_alloc(g->n, ORC_CELL, 1, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        int on = ns_covered(s, i, j, k);
        if (on && excl_fine && s->fine && fine_covers(s->fine, i, j, k)) on = 0;
        A4(&m, i, j, k, 0) = on ? A4(sig, i, j, k, 0) : 0.0;
    }
    orc_sigma_fill_bc(g, &m);
    return m;
}
/* 1 on the cells of the level (not under the next finer level if excl_fine), 0 elsewhere */
static orc_fab cell_mask(const orc_ns_state* s, int excl_fine)
{
    const orc_geom* g = &s->g;
    orc_fab m = orc_alloc(g->n, ORC_CELL, 0, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        int on = ns_covered(s, i, j, k);
        if (on && excl_fine && s->fine && fine_covers(s->fine, i, j, k)) on = 0;
        A4(&m, i, j, k, 0) = on ? 1.0 : 0.0;
    }
    return m;
}
/* velocity (3 comps, 1 ghost) with the same restriction; ghost cells: periodic images of the masked field, cells outside walls
 * keep the incoming values (inflow data; orc_nodal_divu_bc ignores the rest) */
static orc_fab masked_vel(const orc_ns_state* s, const orc_fab* vel, int excl_fine)
{
    const orc_geom* g = &s->g;
    orc_fab m = orc_alloc(g->n, ORC_CELL, 1, 3);
    for (int n = 0; n < 3; ++n)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
        if (!cell_in_domain(g, i, j, k)) { A4(&m, i, j, k, n) = A4(vel, i, j, k, n); continue; }
        int on = ns_covered(s, i, j, k);
        if (on && excl_fine && s->fine && fine_covers(s->fine, i, j, k)) on = 0;
        const int q[3] = {g->periodic[0] ? wrapi(i, g->n[0]) : i, g->periodic[1] ? wrapi(j, g->n[1]) : j, g->periodic[2] ? wrapi(k, g->n[2]) : k};
        A4(&m, i, j, k, n) = on ? A4(vel, q[0], q[1], q[2], n) : 0.0;
    }
    return m;
}

/* full-weighting restriction (1,2,1)^3/64 of a fine nodal field to the nodes of the coarse level; the fine field gets its ghost
 * nodes filled first (periodic images, even reflection about every non-periodic face).  This is the transpose of the trilinear
 * interpolation scaled by 1/ratio^3, i.e. exactly the weights of SyncRegister::FineAdd (SyncRegister.cpp:478-536) */
static void restrict_nodes(const orc_ns_state* f, orc_fab* crse, orc_fab* fine /*1 ghost*/)
{
    static const int NEU3[3] = {ORC_LO_NEUMANN, ORC_LO_NEUMANN, ORC_LO_NEUMANN};
    orc_nodal_fill_bc(&f->g, fine, NEU3, NEU3);
    orc_nodal_restrict(crse, fine, &f->crse->g);
}

/* Hydro::NodalProjector::computeSyncResidualCoarse -> MLNodeLaplacian::compSyncResidualCoarse: on the nodes of level s that touch
 * both cells covered by the finer level and cells that are not, the residual rhs - L(phi) formed with the uncovered cells only
 * (velocity and sigma zeroed under the fine level); zero elsewhere. */
orc_fab amr_sync_resid_crse(const orc_ns_state* s, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig, const orc_fab* rhcc)
{
    const orc_geom* g = &s->g;
    orc_fab r = orc_alloc(g->n, ORC_NODE, 1, 1);
    orc_fab um = masked_vel(s, vold, 1), sm = masked_sigma(s, sig, 1);
    orc_fab rhs = orc_alloc(g->n, ORC_NODE, 0, 1), ax = orc_alloc(g->n, ORC_NODE, 0, 1);
    orc_fab ph = orc_alloc(g->n, ORC_NODE, 1, 1);
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) A4(&ph, i, j, k, 0) = A4(phi, i, j, k, 0);
    orc_nodal_fill_bc(g, &ph, s->nlobc, s->nhibc);
    orc_nodal_divu_bc(g, &rhs, &um, s->nlobc, s->nhibc);
    if (rhcc) { orc_fab cm = cell_mask(s, 1); orc_nodal_rhcc_add(g, &rhs, rhcc, s->nlobc, s->nhibc, &cm); orc_free(&cm); }
    orc_nodal_adotx(g, &ax, &ph, &sm);
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
        A4(&r, i, j, k, 0) = (node_vs_fine(s->fine, i, j, k) == 2 && !on_dirichlet_face(s, i, j, k)) ? A4(&rhs, i, j, k, 0) - A4(&ax, i, j, k, 0) : 0.0;
    orc_free(&um); orc_free(&sm); orc_free(&rhs); orc_free(&ax); orc_free(&ph);
    return r;
}

/* computeSyncResidualFine -> compSyncResidualFine: on the nodes of the boundary of level s (> 0) inside the domain, the residual
 * formed with the cells of the level only; zero elsewhere. */
orc_fab amr_sync_resid_fine(const orc_ns_state* s, const orc_fab* vold, const orc_fab* phi, const orc_fab* sig, const orc_fab* rhcc)
{
    const orc_geom* g = &s->g;
    orc_fab r = orc_alloc(g->n, ORC_NODE, 1, 1);
    orc_fab um = masked_vel(s, vold, 0), sm = masked_sigma(s, sig, 0);
    orc_fab rhs = orc_alloc(g->n, ORC_NODE, 0, 1), ax = orc_alloc(g->n, ORC_NODE, 0, 1);
    orc_fab ph = orc_alloc(g->n, ORC_NODE, 1, 1);
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) A4(&ph, i, j, k, 0) = A4(phi, i, j, k, 0);
    orc_nodal_fill_bc(g, &ph, s->nlobc, s->nhibc);
    orc_nodal_divu_bc(g, &rhs, &um, s->nlobc, s->nhibc);
    if (rhcc) { orc_fab cm = cell_mask(s, 0); orc_nodal_rhcc_add(g, &rhs, rhcc, s->nlobc, s->nhibc, &cm); orc_free(&cm); }
    orc_nodal_adotx(g, &ax, &ph, &sm);
    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
        A4(&r, i, j, k, 0) = (node_class(s, i, j, k) == ND_BOUNDARY && !on_dirichlet_face(s, i, j, k)) ? A4(&rhs, i, j, k, 0) - A4(&ax, i, j, k, 0) : 0.0;
    orc_free(&um); orc_free(&sm); orc_free(&rhs); orc_free(&ax); orc_free(&ph);
    return r;
}

/* ------------------------------------------------------------------------------------------------------------------------
 * SyncRegister (SyncRegister.cpp): nodal values on the faces of the coarsened fine boxes, kept as one single-valued array on the
 * coarse level's nodes. */
static int on_register(const orc_ns_state* f, int i, int j, int k)      /* node of f->crse on a face of a coarsened fine box (or image) */
{
    const orc_geom* cg = &f->crse->g;
    const int r = f->ratio, c[3] = {i, j, k};
    for (int b = 0; b < f->nbox; ++b) {
        const int* bx = f->boxes + 6 * b;
        for (int sz = -1; sz <= 1; ++sz) for (int sy = -1; sy <= 1; ++sy) for (int sx = -1; sx <= 1; ++sx) {
            const int sh[3] = {sx, sy, sz};
            int ok = 1, onface = 0;
            for (int d = 0; d < 3 && ok; ++d) {
                if (sh[d] != 0 && !cg->periodic[d]) { ok = 0; break; }
                const int lo = bx[d] / r + sh[d] * cg->n[d], hi = (bx[3 + d] + 1) / r + sh[d] * cg->n[d];   /* nodal box of the coarsened box */
                if (c[d] < lo || c[d] > hi) ok = 0;
                else if (c[d] == lo || c[d] == hi) onface = 1;
            }
            if (ok && onface) return 1;
        }
    }
    return 0;
}
void syncreg_crse_init(orc_ns_state* f, const orc_fab* resid_crse, double mult)
{
    const orc_geom* cg = &f->crse->g;
    for (int k = 0; k <= cg->n[2]; ++k) for (int j = 0; j <= cg->n[1]; ++j) for (int i = 0; i <= cg->n[0]; ++i)
        A4(&f->sync_reg, i, j, k, 0) = on_register(f, i, j, k) ? mult * A4(resid_crse, i, j, k, 0) : 0.0;
}
void syncreg_fine_add(orc_ns_state* f, const orc_fab* resid_fine, double mult)
{
    const orc_geom* cg = &f->crse->g;
    orc_fab rf = orc_alloc(f->g.n, ORC_NODE, 1, 1), rc = orc_alloc(cg->n, ORC_NODE, 0, 1);
    for (int k = 0; k <= f->g.n[2]; ++k) for (int j = 0; j <= f->g.n[1]; ++j) for (int i = 0; i <= f->g.n[0]; ++i) A4(&rf, i, j, k, 0) = mult * A4(resid_fine, i, j, k, 0);
    restrict_nodes(f, &rc, &rf);
    for (int k = 0; k <= cg->n[2]; ++k) for (int j = 0; j <= cg->n[1]; ++j) for (int i = 0; i <= cg->n[0]; ++i)
        if (on_register(f, i, j, k)) A4(&f->sync_reg, i, j, k, 0) += A4(&rc, i, j, k, 0);
    orc_free(&rf); orc_free(&rc);
}
/* SyncRegister::InitRHS (SyncRegister.cpp:47-304): rhs = register values; zero on outflow faces; zero on the nodes that are
 * surrounded by fine cells only (bndry_mask) */
static void syncreg_init_rhs(const orc_ns_state* f, orc_fab* rhs)
{
    const orc_ns_state* c = f->crse;
    const orc_geom* cg = &c->g;
    for (int k = 0; k <= cg->n[2]; ++k) for (int j = 0; j <= cg->n[1]; ++j) for (int i = 0; i <= cg->n[0]; ++i) {
        double v = on_register(f, i, j, k) ? A4(&f->sync_reg, i, j, k, 0) : 0.0;
        const int idx[3] = {i, j, k};
        for (int d = 0; d < 3; ++d) {
            if (cg->periodic[d]) continue;
            if (idx[d] == 0 && c->p.phys_lo[d] == 2) v = 0.0;             /* PhysBCType::outflow */
            if (idx[d] == cg->n[d] && c->p.phys_hi[d] == 2) v = 0.0;
        }
        if (node_vs_fine(f, i, j, k) == 1) v = 0.0;
        A4(rhs, i, j, k, 0) = v;
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * Composite nodal projection over levels lev[c0 .. c0+nl-1] (Hydro::NodalProjector::project on several AMR levels as driven by
 * Projection::doMLMGNodalProjection, Projection.cpp:2385-2567).
 *
 * Discretisation: the conforming Q1 finite-element composite operator.  Every cell of a level that is not covered by the next
 * finer level contributes its element matrix (sigma of that level); a node of level l+1 on the boundary of that level is a slave:
 * its value is the trilinear interpolant of the level-l nodes, and what the fine cells contribute to it is handed to those
 * level-l nodes with the transposed weights / ratio^3 (the weights of SyncRegister::FineAdd).  Nodes of the coarsest level on its
 * own boundary (c0 > 0) and nodes on Dirichlet (outflow) faces keep their incoming value.  The right-hand side is assembled the
 * same way from div(vel) of the uncovered cells (+ rhnd on the coarsest level).  Solved with conjugate gradients in the inner
 * product that makes the finite-difference-scaled operator self-adjoint (weights ratio^-3l, 1/2 per Neumann wall).
 * After the solve: vel -= sigma grad(phi) on every cell of every level, Gradp = / += grad(phi), phi of covered coarse nodes =
 * injection of the fine phi, velocities averaged down (NodalProjector::averageDown). */
typedef struct clev {
    orc_ns_state* s;
    orc_fab sigm;      /* sigma on the uncovered cells of the level, ghost cells filled */
    orc_fab own;       /* node: 1 = unknown of the composite system */
    orc_fab slave;     /* node: 1 = slave of the next coarser level */
    orc_fab wt;        /* node weight in inner products */
} clev;

static void comp_fill_slaves(clev* L, int nl, orc_fab* x /*[nl], node, 1 ghost*/)
{
    for (int l = 0; l < nl; ++l) {
        const orc_geom* g = &L[l].s->g;
        if (l > 0) {
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
                if (A4(&L[l].slave, i, j, k, 0) == 0.0) continue;
                const int r = L[l].s->ratio;
                const int f[3] = {i, j, k};
                int c0[3]; double w[3];
                for (int d = 0; d < 3; ++d) { c0[d] = f[d] / r; w[d] = (double)(f[d] - c0[d] * r) / (double)r; }
                double v = 0.0;
                for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
                    const double ww = (cx ? w[0] : 1.0 - w[0]) * (cy ? w[1] : 1.0 - w[1]) * (cz ? w[2] : 1.0 - w[2]);
                    if (ww != 0.0) v += ww * A4(&x[l - 1], c0[0] + cx, c0[1] + cy, c0[2] + cz, 0);
                }
                A4(&x[l], i, j, k, 0) = v;
            }
        }
        orc_nodal_fill_bc(g, &x[l], L[l].s->nlobc, L[l].s->nhibc);
    }
}

/* y = A x on the unknowns (0 elsewhere); x must have its slaves filled */
static void comp_apply(clev* L, int nl, orc_fab* y, orc_fab* x)
{
    comp_fill_slaves(L, nl, x);
    orc_fab carry; carry.p = NULL;          /* restricted boundary contributions of the next finer level */
    for (int l = nl - 1; l >= 0; --l) {
        const orc_geom* g = &L[l].s->g;
        orc_nodal_adotx(g, &y[l], &x[l], &L[l].sigm);
        if (carry.p) {
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) A4(&y[l], i, j, k, 0) += A4(&carry, i, j, k, 0);
            orc_free(&carry); carry.p = NULL;
        }
        if (l > 0) {
            orc_fab b = orc_alloc(g->n, ORC_NODE, 1, 1);
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                A4(&b, i, j, k, 0) = A4(&L[l].slave, i, j, k, 0) != 0.0 ? A4(&y[l], i, j, k, 0) : 0.0;
            carry = orc_alloc(L[l - 1].s->g.n, ORC_NODE, 0, 1);
            restrict_nodes(L[l].s, &carry, &b);
            orc_free(&b);
        }
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
            if (A4(&L[l].own, i, j, k, 0) == 0.0) A4(&y[l], i, j, k, 0) = 0.0;
    }
}
static double comp_dot(clev* L, int nl, orc_fab* a, orc_fab* b)
{
    double s = 0.0;
    for (int l = 0; l < nl; ++l) {
        const orc_geom* g = &L[l].s->g;
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
            const double w = A4(&L[l].wt, i, j, k, 0);
            if (w != 0.0) s += w * (A4(&a[l], i, j, k, 0) * A4(&b[l], i, j, k, 0));
        }
    }
    return s;
}
static double comp_norminf(clev* L, int nl, orc_fab* a)
{
    double m = 0.0;
    for (int l = 0; l < nl; ++l) {
        const orc_geom* g = &L[l].s->g;
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
            if (A4(&L[l].own, i, j, k, 0) != 0.0) { const double v = fabs(A4(&a[l], i, j, k, 0)); if (v > m) m = v; }
    }
    return m;
}
static void comp_axpy(clev* L, int nl, orc_fab* y, double a, orc_fab* x)       /* y += a x on the unknowns */
{
    for (int l = 0; l < nl; ++l) {
        const orc_geom* g = &L[l].s->g;
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
            if (A4(&L[l].own, i, j, k, 0) != 0.0) A4(&y[l], i, j, k, 0) += a * A4(&x[l], i, j, k, 0);
    }
}

void amr_composite_project(orc_amr* a, int c0, int nl, orc_fab* vel[] /*cell, 3 comps, 1 ghost*/, orc_fab* phi[] /*node, 1 ghost*/,
                           const orc_fab* sig[] /*cell, valid*/, const orc_fab* rhnd /*nodes of level c0 or NULL*/, double rtol, double atol,
                           int increment_gp, double inflow_scale, orc_mg_stats* st)
{
    amr_composite_project_rhcc(a, c0, nl, vel, phi, sig, rhnd, NULL, rtol, atol, increment_gp, inflow_scale, st);
}
/* the same with a cell-centred source per level (rhcc[l] or NULL): div(sig grad phi) = div(vel) + rhnd + <rhcc>, <.> = the node average
 * over the uncovered cells of the level (MLNodeLaplacian::compRHS) */
void amr_composite_project_rhcc(orc_amr* a, int c0, int nl, orc_fab* vel[], orc_fab* phi[], const orc_fab* sig[], const orc_fab* rhnd,
                                orc_fab* const rhcc[], double rtol, double atol, int increment_gp, double inflow_scale, orc_mg_stats* st)
{
    clev L[8];
    orc_fab b[8], x[8], r[8], p[8], q[8];
    int singular = 1;
    double scale = 1.0;
    for (int l = 0; l < nl; ++l) {
        orc_ns_state* s = a->lev[c0 + l];
        const orc_geom* g = &s->g;
        const int has_fine = l < nl - 1;
        L[l].s = s;
        /* sigma on the cells of the level not covered by the next level OF THIS SOLVE */
        orc_fab m = orc_alloc(g->n, ORC_CELL, 1, 1);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
            int on = ns_covered(s, i, j, k);
            if (on && has_fine && fine_covers(s->fine, i, j, k)) on = 0;
            A4(&m, i, j, k, 0) = on ? A4(sig[l], i, j, k, 0) : 0.0;
        }
        orc_sigma_fill_bc(g, &m);
        L[l].sigm = m;
        L[l].own = orc_alloc(g->n, ORC_NODE, 0, 1); L[l].slave = orc_alloc(g->n, ORC_NODE, 0, 1); L[l].wt = orc_alloc(g->n, ORC_NODE, 0, 1);
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
            const int cls = node_class(s, i, j, k);
            int own = cls == ND_INTERIOR, slave = 0;
            if (cls == ND_BOUNDARY) { if (l > 0) slave = 1; else singular = 0; }      /* boundary of the coarsest level: Dirichlet data */
            if (own && on_dirichlet_face(s, i, j, k)) { own = 0; singular = 0; }
            if (own && has_fine && node_vs_fine(s->fine, i, j, k) == 1) own = 0;
            A4(&L[l].own, i, j, k, 0) = own; A4(&L[l].slave, i, j, k, 0) = slave;
            A4(&L[l].wt, i, j, k, 0) = own ? node_wt(s, i, j, k) * scale : 0.0;
        }
        scale /= (double)(s->fine ? s->fine->ratio * s->fine->ratio * s->fine->ratio : 8);
        b[l] = orc_alloc(g->n, ORC_NODE, 1, 1); x[l] = orc_alloc(g->n, ORC_NODE, 1, 1); r[l] = orc_alloc(g->n, ORC_NODE, 1, 1);
        p[l] = orc_alloc(g->n, ORC_NODE, 1, 1); q[l] = orc_alloc(g->n, ORC_NODE, 1, 1);
    }
    /* right-hand side: div(vel) of the uncovered cells (+ rhnd), fine boundary contributions handed down */
    {
        orc_fab carry; carry.p = NULL;
        for (int l = nl - 1; l >= 0; --l) {
            orc_ns_state* s = L[l].s;
            const orc_geom* g = &s->g;
            orc_fill_periodic(vel[l], g, ORC_CELL);
            ns_set_inflow_ghosts(s, vel[l], inflow_scale);
            orc_fab um = orc_alloc(g->n, ORC_CELL, 1, 3);
            for (int n = 0; n < 3; ++n)
            for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
                if (!cell_in_domain(g, i, j, k)) { A4(&um, i, j, k, n) = A4(vel[l], i, j, k, n); continue; }
                int on = ns_covered(s, i, j, k);
                if (on && l < nl - 1 && fine_covers(s->fine, i, j, k)) on = 0;
                A4(&um, i, j, k, n) = on ? A4(vel[l], i, j, k, n) : 0.0;
            }
            orc_fab d = orc_alloc(g->n, ORC_NODE, 0, 1);
            orc_nodal_divu_bc(g, &d, &um, s->nlobc, s->nhibc);
            if (rhcc && rhcc[l]) { orc_fab cm = cell_mask(s, l < nl - 1); orc_nodal_rhcc_add(g, &d, rhcc[l], s->nlobc, s->nhibc, &cm); orc_free(&cm); }
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
                double v = A4(&d, i, j, k, 0);
                if (l == 0 && rhnd) v += A4(rhnd, i, j, k, 0);
                if (carry.p) v += A4(&carry, i, j, k, 0);
                A4(&b[l], i, j, k, 0) = v;
            }
            if (carry.p) { orc_free(&carry); carry.p = NULL; }
            if (l > 0) {
                orc_fab bb = orc_alloc(g->n, ORC_NODE, 1, 1);
                for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                    A4(&bb, i, j, k, 0) = A4(&L[l].slave, i, j, k, 0) != 0.0 ? A4(&b[l], i, j, k, 0) : 0.0;
                carry = orc_alloc(L[l - 1].s->g.n, ORC_NODE, 0, 1);
                restrict_nodes(s, &carry, &bb);
                orc_free(&bb);
            }
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                if (A4(&L[l].own, i, j, k, 0) == 0.0) A4(&b[l], i, j, k, 0) = 0.0;
            orc_free(&um); orc_free(&d);
        }
    }
    /* incoming phi: Dirichlet data on the non-unknown nodes, initial guess elsewhere; move the data to the right-hand side */
    for (int l = 0; l < nl; ++l) {
        const orc_geom* g = &L[l].s->g;
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) A4(&x[l], i, j, k, 0) = A4(phi[l], i, j, k, 0);
    }
    if (singular) {                      /* MLMG::makeSolvable: remove the mean of the right-hand side over the composite unknowns */
        double sw = 0.0, sb = 0.0;
        for (int l = 0; l < nl; ++l) {
            const orc_geom* g = &L[l].s->g;
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
                const double w = A4(&L[l].wt, i, j, k, 0);
                sw += w; sb += w * A4(&b[l], i, j, k, 0);
            }
        }
        const double off = sb / sw;
        for (int l = 0; l < nl; ++l) {
            const orc_geom* g = &L[l].s->g;
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                if (A4(&L[l].own, i, j, k, 0) != 0.0) A4(&b[l], i, j, k, 0) -= off;
        }
    }
    comp_apply(L, nl, q, x);
    for (int l = 0; l < nl; ++l) {
        const orc_geom* g = &L[l].s->g;
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
            A4(&r[l], i, j, k, 0) = A4(&L[l].own, i, j, k, 0) != 0.0 ? A4(&b[l], i, j, k, 0) - A4(&q[l], i, j, k, 0) : 0.0;
            A4(&p[l], i, j, k, 0) = A4(&r[l], i, j, k, 0);
        }
    }
    orc_mg_stats loc; memset(&loc, 0, sizeof(loc));
    loc.rhsnorm0 = comp_norminf(L, nl, b); loc.resnorm0 = comp_norminf(L, nl, r); loc.resnorm = loc.resnorm0;
    const double max_norm = loc.rhsnorm0 >= loc.resnorm0 ? loc.rhsnorm0 : loc.resnorm0;
    const double target = fmax(atol, fmax(rtol, 1.e-16) * max_norm);
    double rr = comp_dot(L, nl, r, r);
    if (loc.resnorm0 <= target) loc.converged = 1;
    for (int it = 0; it < 20000 && !loc.converged; ++it) {
        /* the search direction lives on the unknowns; slaves / Dirichlet nodes of a direction are zero / interpolated inside comp_apply */
        for (int l = 0; l < nl; ++l) {
            const orc_geom* g = &L[l].s->g;
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                if (A4(&L[l].own, i, j, k, 0) == 0.0) A4(&p[l], i, j, k, 0) = 0.0;
        }
        comp_apply(L, nl, q, p);
        const double pq = comp_dot(L, nl, p, q);
        if (pq == 0.0) break;
        const double alpha = rr / pq;
        comp_axpy(L, nl, x, alpha, p);
        comp_axpy(L, nl, r, -alpha, q);
        loc.resnorm = comp_norminf(L, nl, r);
        loc.iters = it + 1;
        if (loc.resnorm <= target) { loc.converged = 1; break; }
        const double rr1 = comp_dot(L, nl, r, r);
        const double beta = rr1 / rr;
        rr = rr1;
        for (int l = 0; l < nl; ++l) {
            const orc_geom* g = &L[l].s->g;
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                A4(&p[l], i, j, k, 0) = A4(&r[l], i, j, k, 0) + beta * A4(&p[l], i, j, k, 0);
        }
    }
    if (!loc.converged) fprintf(stderr, "orc composite nodal solve: not converged (res %.3e target %.3e)\n", loc.resnorm, target);
    if (singular) {                      /* the solution of the singular system is fixed by a zero weighted mean over the composite unknowns */
        double sw = 0.0, sx = 0.0;
        for (int l = 0; l < nl; ++l) {
            const orc_geom* g = &L[l].s->g;
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
                const double w = A4(&L[l].wt, i, j, k, 0);
                sw += w; sx += w * A4(&x[l], i, j, k, 0);
            }
        }
        const double off = sx / sw;
        for (int l = 0; l < nl; ++l) {
            const orc_geom* g = &L[l].s->g;
            for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                if (A4(&L[l].own, i, j, k, 0) != 0.0) A4(&x[l], i, j, k, 0) -= off;
        }
    }
    if (st) *st = loc;
    /* slaves, covered coarse nodes (injection of the fine solution), ghost nodes */
    comp_fill_slaves(L, nl, x);
    for (int l = nl - 2; l >= 0; --l) {
        const orc_geom* g = &L[l].s->g;
        const orc_ns_state* f = L[l + 1].s;
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
            if (node_vs_fine(f, i, j, k) != 0) A4(&x[l], i, j, k, 0) = A4(&x[l + 1], f->ratio * i, f->ratio * j, f->ratio * k, 0);
        orc_nodal_fill_bc(g, &x[l], L[l].s->nlobc, L[l].s->nhibc);
    }
    for (int l = 0; l < nl; ++l) {
        orc_ns_state* s = L[l].s;
        const orc_geom* g = &s->g;
        orc_copy_all(phi[l], &x[l]);
        /* vel -= sigma grad phi, Gradp = / += grad phi on the cells of the level */
        orc_fab gp = orc_alloc(g->n, ORC_CELL, 0, 3);
        orc_nodal_compgrad(g, &gp, phi[l]);
        orc_fab* G = GP_NEW(s);
        for (int n = 0; n < 3; ++n)
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
            if (!ns_covered(s, i, j, k)) continue;
            A4(vel[l], i, j, k, n) -= A4(sig[l], i, j, k, 0) * A4(&gp, i, j, k, n);
            if (increment_gp) A4(G, i, j, k, n) += A4(&gp, i, j, k, n); else A4(G, i, j, k, n) = A4(&gp, i, j, k, n);
        }
        orc_free(&gp);
    }
    /* NodalProjector::averageDown(vel) */
    for (int l = nl - 1; l >= 1; --l) {
        orc_ns_state* f = L[l].s;
        const orc_geom* cg = &L[l - 1].s->g;
        const int rr_ = f->ratio;
        for (int n = 0; n < 3; ++n)
        for (int k = 0; k < cg->n[2]; ++k) for (int j = 0; j < cg->n[1]; ++j) for (int i = 0; i < cg->n[0]; ++i) {
            if (!fine_covers(f, i, j, k)) continue;
            double sm = 0.0;
            for (int c = 0; c < rr_; ++c) for (int bq = 0; bq < rr_; ++bq) for (int aq = 0; aq < rr_; ++aq) sm += A4(vel[l], rr_ * i + aq, rr_ * j + bq, rr_ * k + c, n);
            A4(vel[l - 1], i, j, k, n) = sm / (double)(rr_ * rr_ * rr_);
        }
    }
    for (int l = 0; l < nl; ++l) {
        orc_ns_state* s = L[l].s;
        ns_fill_gp(s, GP_NEW(s), 0.5 * (s->pt_new[0] + s->pt_new[1]));
        orc_free(&L[l].sigm); orc_free(&L[l].own); orc_free(&L[l].slave); orc_free(&L[l].wt);
        orc_free(&b[l]); orc_free(&x[l]); orc_free(&r[l]); orc_free(&p[l]); orc_free(&q[l]);
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * amrex::average_down (cells), average_down_nodal (injection) */
static void avg_down_cells(const orc_ns_state* f, const orc_fab* fine, orc_fab* crse, int sc, int nc)
{
    const orc_geom* cg = &f->crse->g;
    const int r = f->ratio;
    for (int n = 0; n < nc; ++n)
    for (int k = 0; k < cg->n[2]; ++k) for (int j = 0; j < cg->n[1]; ++j) for (int i = 0; i < cg->n[0]; ++i) {
        if (!fine_covers(f, i, j, k)) continue;
        double s = 0.0;
        for (int c = 0; c < r; ++c) for (int b = 0; b < r; ++b) for (int a = 0; a < r; ++a) s += A4(fine, r * i + a, r * j + b, r * k + c, sc + n);
        A4(crse, i, j, k, sc + n) = s * (1.0 / (double)(r * r * r));
    }
}
static void avg_down_nodes(const orc_ns_state* f, const orc_fab* fine, orc_fab* crse)
{
    const orc_geom* cg = &f->crse->g;
    const int r = f->ratio;
    for (int k = 0; k <= cg->n[2]; ++k) for (int j = 0; j <= cg->n[1]; ++j) for (int i = 0; i <= cg->n[0]; ++i)
        if (node_vs_fine(f, i, j, k) != 0) A4(crse, i, j, k, 0) = A4(fine, r * i, r * j, r * k, 0);
}

/* NavierStokes::avgDown (NavierStokes.cpp:1845-1873) + avgDown_StatePress (NavierStokesBase.cpp:4125-4163) */
static void avg_down(orc_amr* a, int lev)
{
    orc_ns_state *c = a->lev[lev], *f = a->lev[lev + 1];
    avg_down_cells(f, S_NEW(f), S_NEW(c), 0, c->nalloc);        /* state, and divu / dsdt (NavierStokes.cpp:1859-1872) */
    for (int l = lev; l < a->nlev; ++l) ns_make_rho_curr_time(a->lev[l]);
    avg_down_nodes(f, c->initial_step ? P_NEW(f) : &f->p_avg, P_NEW(c));
    avg_down_cells(f, GP_NEW(f), GP_NEW(c), 0, 3);
    /* The reference leaves the ghost cells of the coarse Gradp as they were (filled before the average), which makes the next
     * predictor depend on how the coarse level happens to be chopped into boxes.  Here (and in the product) they are re-filled. */
    ns_fill_gp(c, GP_NEW(c), 0.5 * (c->pt_new[0] + c->pt_new[1]));
}

/* NavierStokes::reflux (NavierStokes.cpp:1736-1838) */
static void reflux(orc_amr* a, int lev)
{
    orc_ns_state *c = a->lev[lev], *f = a->lev[lev + 1];
    const orc_geom* g = &c->g;
    const double vol = g->dx[0] * g->dx[1] * g->dx[2], dt_crse = a->dt_level[lev];
    reg_reflux(f, f->reg_visc, &c->Vsync, vol, 1.0, 0, 0, 3);
    reg_reflux(f, f->reg_visc, &c->Ssync, vol, 1.0, 3, 0, c->nstate - 3);
    if (c->p.do_mom_diff == 0)
        for (int n = 0; n < 3; ++n)
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&c->Vsync, i, j, k, n) /= A4(&c->rho_half, i, j, k, 0);
    for (int istate = 3; istate < c->nstate; ++istate) {
        const int conservative = c->scal_cons[istate - Density];
        if (!conservative)
            for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&c->Ssync, i, j, k, istate - 3) /= A4(&c->rho_half, i, j, k, 0);
    }
    reg_reflux(f, f->reg_adv, &c->Vsync, vol, 1.0, 0, 0, 3);
    reg_reflux(f, f->reg_adv, &c->Ssync, vol, 1.0, 3, 0, c->nstate - 3);
    const double scale = 1.0 / dt_crse;
    { size_t N = orc_npts(&c->Vsync) * 3; for (size_t q = 0; q < N; ++q) c->Vsync.p[q] *= scale; }
    { size_t N = orc_npts(&c->Ssync) * (c->nstate - 3); for (size_t q = 0; q < N; ++q) c->Ssync.p[q] *= scale; }
    /* zero the coarse cells under the fine grids (grown tile box: ghost cells included) */
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
        if (!cell_in_domain(g, i, j, k) || !fine_covers(f, i, j, k)) continue;
        for (int n = 0; n < 3; ++n) A4(&c->Vsync, i, j, k, n) = 0.0;
        for (int n = 0; n < c->nstate - 3; ++n) A4(&c->Ssync, i, j, k, n) = 0.0;
    }
}

/* conservative-linear interpolation of a coarse cell field (valid data; ghost cells built here: periodic images, homogeneous
 * ext_dir / extrapolation outside walls) to the cells of the fine level: NavierStokesBase::SyncInterp with cell_cons_interp */
static orc_fab sync_interp(const orc_ns_state* c, const orc_ns_state* f, const orc_fab* crse, int sc, int nc, const orc_bcrec* bc)
{
    int ratio = 1;
    for (const orc_ns_state* q = f; q != c; q = q->crse) ratio *= q->ratio;
    const orc_geom* cg = &c->g;
    orc_fab cd = orc_alloc(cg->n, ORC_CELL, 2, nc);
    for (int n = 0; n < nc; ++n)
    for (int k = 0; k < cg->n[2]; ++k) for (int j = 0; j < cg->n[1]; ++j) for (int i = 0; i < cg->n[0]; ++i) A4(&cd, i, j, k, n) = A4(crse, i, j, k, sc + n);
    orc_fill_periodic(&cd, cg, ORC_CELL);
    double zero[24]; memset(zero, 0, sizeof(zero));
    orc_fill_physbc_cc(&cd, cg, bc, zero, zero);                  /* HomExtDirFill */
    orc_fab fd = orc_alloc(f->g.n, ORC_CELL, 0, nc);
    const int cdomlo[3] = {0, 0, 0}, cdomhi[3] = {cg->n[0] - 1, cg->n[1] - 1, cg->n[2] - 1};
    const int vlo[3] = {1, 1, 1}, vhi[3] = {0, 0, 0};
    orc_fill_coarse_fine(&fd, fd.lo, fd.hi, vlo, vhi, &cd, cdomlo, cdomhi, cg->periodic, ratio, bc);
    orc_free(&cd);
    return fd;
}

/* MacProj::mac_sync_solve (MacProj.cpp:359-479) */
static void mac_sync_solve(orc_amr* a, int lev, orc_fab Ucorr[3])
{
    orc_ns_state *c = a->lev[lev], *f = a->lev[lev + 1];
    const orc_geom* g = &c->g;
    const double dt = a->dt_level[lev];
    orc_fab Rhs = orc_alloc(g->n, ORC_CELL, 0, 1);
    reg_reflux(f, f->reg_mac, &Rhs, g->dx[0] * g->dx[1] * g->dx[2], -1.0, 0, 0, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
        if (fine_covers(f, i, j, k)) A4(&Rhs, i, j, k, 0) = 0.0;
        A4(&Rhs, i, j, k, 0) = -A4(&Rhs, i, j, k, 0);             /* Rhs.negate() */
    }
    orc_setval(&c->mac_phi, 0.0);                                  /* mac_sync_phi re-uses mac_phi_crse[level] */
    orc_fab* um[3];
    for (int d = 0; d < 3; ++d) { Ucorr[d] = orc_alloc(g->n, ORC_FACE[d], 1, 1); um[d] = &Ucorr[d]; }
    orc_mg_opts o = c->o; o.maxorder = 4;
    orc_mg_stats st;
    /* mlmg_mac_solve with a null velocity: rhs = S, solve, the returned fluxes -b grad(phi) are the correction; IAMR negates them.
     * On a refined level cphi is null (MacProj.cpp:454-456): homogeneous Dirichlet data on the coarse/fine faces */
    if (lev == 0) orc_mac_project(g, um, &c->rho_half, &Rhs, &c->mac_phi, 2.0 / dt, c->lobc, c->hibc, 1.e-10 /*mac_sync_tol, MacProj.cpp:44*/, c->p.mac_abs_tol, &o, &st);
    else {
        orc_fab zero = orc_alloc(c->crse->g.n, ORC_CELL, 1, 1);
        orc_mac_project_cf(g, um, &c->rho_half, &Rhs, &c->mac_phi, 2.0 / dt, c->lobc, c->hibc, c->nbox, c->boxes, c->ratio, &zero, 1.e-10, c->p.mac_abs_tol, &o, &st);
        orc_free(&zero);
    }
    for (int d = 0; d < 3; ++d) {
        const size_t N = orc_npts(&Ucorr[d]);
        for (size_t q = 0; q < N; ++q) Ucorr[d].p[q] = -Ucorr[d].p[q];
        orc_fill_periodic(&Ucorr[d], g, ORC_FACE[d]);
    }
    orc_free(&Rhs);
}

/* MacProj::mac_sync_compute (MacProj.cpp:490-731), inviscid / level-0-viscous form: re-advection of the state with Ucorr */
void orc_compute_aofs_sync(const orc_geom* g, orc_fab* sync, int acomp, const orc_fab* S, int ncomp,
                           const orc_fab* force, const orc_fab* divu, orc_fab* const umac[3], orc_fab* const ucorr[3], const int* iconserv,
                           double dt, const orc_bcrec* bc, int is_velocity, int use_forces_in_trans, orc_fab* flux_out[3]);
static void mac_sync_compute(orc_amr* a, int lev, orc_fab Ucorr[3])
{
    orc_ns_state* c = a->lev[lev];
    const orc_geom* g = &c->g;
    const double dt = a->dt_level[lev], prev_time = c->st_old;
    orc_godunov_set_ppm(c->p.use_ppm);
    orc_fab Smf = ns_fillpatch_time(c, prev_time, 0, 0, 3, 3);
    orc_fab Sc = ns_fillpatch_time(c, prev_time, 0, Density, c->nscal, 3);
    const int mom = c->p.do_mom_diff;
    if (mom) { const size_t N = orc_npts(&Smf); for (int n = 0; n < 3; ++n) for (size_t q = 0; q < N; ++q) Smf.p[q + N * n] *= Sc.p[q]; }
    orc_fab tfv = orc_alloc(g->n, ORC_CELL, 1, 3), tfs = orc_alloc(g->n, ORC_CELL, 1, c->nscal), divu = ns_divu_half(c, dt, 1, 0);   /* getDivCond(nghost_force, prev_time), MacProj.cpp:562 */
    const orc_fab* Gp = GP_OLD(c);
    /* viscous forcing at the old time (MacProj.cpp:566-572) */
    orc_fab vvisc = orc_alloc(g->n, ORC_CELL, 1, 3);
    if (c->p.be_cn_theta != 1.0 && c->p.visc_coef > 0.0) ns_get_visc_terms_vel(c, &vvisc, S_OLD(c));
    for (int n = 0; n < 3; ++n)
    for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) {
        const double rho = A4(&Sc, i, j, k, 0);
        double t = ((fabs(c->p.gravity) > 0.0001 && n == 2) ? c->p.gravity * rho : 0.0) + A4(&vvisc, i, j, k, n) - A4(Gp, i, j, k, n);
        if (!mom) t /= rho;
        A4(&tfv, i, j, k, n) = t;
    }
    orc_free(&vvisc);
    /* scalars: getForce = 0; conservative: tf += visc; convective: tf = tf/rho + visc (MacProj.cpp:641-683); density does not diffuse */
    for (int n = 1; n < c->nscal; ++n) {
        if (!(c->p.be_cn_theta != 1.0 && c->scal_diff[n] > 0.0)) continue;
        orc_fab sv = orc_alloc(g->n, ORC_CELL, 1, 1);
        ns_get_visc_terms_scalar(c, &sv, S_OLD(c), Density + n);
        for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i)
            A4(&tfs, i, j, k, n) = Density + n == c->Temp ? A4(&sv, i, j, k, 0) / A4(&Sc, i, j, k, 0) : A4(&sv, i, j, k, 0);   /* MacProj.cpp:641-683 */
        orc_free(&sv);
    }
    orc_fab *um[3] = {&c->umac[0], &c->umac[1], &c->umac[2]}, *uc[3] = {&Ucorr[0], &Ucorr[1], &Ucorr[2]};
    const int icv[3] = {mom, mom, mom};
    int ics[ORC_MAXSCAL]; for (int n = 0; n < c->nscal; ++n) ics[n] = c->scal_cons[n];
    orc_fab flv[3], fls[3]; orc_fab *flvp[3], *flsp[3];
    for (int d = 0; d < 3; ++d) { flv[d] = orc_alloc(g->n, ORC_FACE[d], 0, 3); fls[d] = orc_alloc(g->n, ORC_FACE[d], 0, c->nscal); flvp[d] = &flv[d]; flsp[d] = &fls[d]; }
    orc_compute_aofs_sync(g, &c->Vsync, 0, &Smf, 3, &tfv, &divu, um, uc, icv, dt, c->bc_vel, 1, c->p.use_forces_in_trans, flvp);
    orc_compute_aofs_sync(g, &c->Ssync, 0, &Sc, c->nscal, &tfs, &divu, um, uc, ics, dt, c->bc_scal, 0, c->p.use_forces_in_trans, flsp);
    /* NavierStokesBase.cpp:5083-5096 with sync_factor = -1 (do_crse_add = false) */
    for (int d = 0; d < 3; ++d) {
        if (c->fine) { reg_crse_init(c->fine, c->fine->reg_adv, &flv[d], d, 0, 0, 3, dt, 1); reg_crse_init(c->fine, c->fine->reg_adv, &fls[d], d, 0, Density, c->nscal, dt, 1); }
        if (c->level > 0) { reg_fine_add(c, c->reg_adv, &flv[d], d, 0, 0, 3, -dt); reg_fine_add(c, c->reg_adv, &fls[d], d, 0, Density, c->nscal, -dt); }
        if (c->level > 0) reg_fine_add(c, c->reg_mac, &Ucorr[d], d, 0, 0, 1, -g->dx[(d + 1) % 3] * g->dx[(d + 2) % 3] / (double)a->n_cycle[lev]);
        orc_free(&flv[d]); orc_free(&fls[d]);
    }
    orc_free(&Smf); orc_free(&Sc); orc_free(&tfv); orc_free(&tfs); orc_free(&divu);
}

/* NavierStokes::mac_sync (NavierStokes.cpp:1438-1730), non-diffusive scalars, inviscid velocity */
static void mac_sync(orc_amr* a, int lev)
{
    orc_ns_state* c = a->lev[lev];
    const orc_geom* g = &c->g;
    const double dt = a->dt_level[lev];
    const int numscal = c->nstate - 3;
    orc_fab Ucorr[3];
    mac_sync_solve(a, lev, Ucorr);
    mac_sync_compute(a, lev, Ucorr);
    for (int d = 0; d < 3; ++d) orc_free(&Ucorr[d]);
    orc_fab* Sn = S_NEW(c);
    orc_fab Delta = orc_alloc(g->n, ORC_CELL, 0, ORC_MAXSCAL);
    for (int sn = 1; sn < numscal; ++sn)        /* :1500-1528: conservative Q = rho q: sync -= (sync of rho) * q */
        if (c->scal_cons[sn])
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
            A4(&Delta, i, j, k, sn) = A4(Sn, i, j, k, Density + sn) * A4(&c->Ssync, i, j, k, 0) / A4(Sn, i, j, k, Density);
            A4(&c->Ssync, i, j, k, sn) -= A4(&Delta, i, j, k, sn);
        }
    if (c->p.do_mom_diff == 1)
        for (int n = 0; n < 3; ++n)
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&c->Vsync, i, j, k, n) /= A4(Sn, i, j, k, Density);
    const double theta = c->p.be_cn_theta;
    if (c->p.visc_coef > 0.0) {
        /* Diffusion::diffuse_Vsync -> diffuse_tensor_Vsync (Diffusion.cpp:960-1178): (rho - theta dt div tau) Vsync' = rho Vsync with
         * homogeneous boundary and coarse/fine data; upstream sets the face coefficients of this solve to 1.0 (:1122-1135), not to the
         * viscosity -- followed as written */
        const int rf3 = c->p.do_mom_diff;
        orc_fab Rhs = orc_alloc(g->n, ORC_CELL, 0, 3), acoef = orc_alloc(g->n, ORC_CELL, 0, 1), Soln = orc_alloc(g->n, ORC_CELL, 1, 3);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
            const double r = rf3 ? A4(S_OLD(c), i, j, k, Density) : A4(&c->rho_half, i, j, k, 0);
            for (int n = 0; n < 3; ++n) A4(&Rhs, i, j, k, n) = A4(&c->Vsync, i, j, k, n) * r;
            A4(&acoef, i, j, k, 0) = rf3 ? A4(Sn, i, j, k, Density) : A4(&c->rho_half, i, j, k, 0);
        }
        orc_fab one[3]; orc_fab* ep[3];
        for (int d = 0; d < 3; ++d) { one[d] = orc_alloc(g->n, ORC_FACE[d], 0, 1); orc_setval(&one[d], 1.0); ep[d] = &one[d]; }
        orc_mg_opts vo = c->o; vo.maxorder = 2;
        orc_mg_stats st;
        if (c->level > 0) orc_tensor_solve_cf(g, c->nbox, c->boxes, c->ratio, &Soln, &Rhs, 1.0, theta * dt, &acoef, ep, c->vlobc, c->vhibc, NULL, c->p.visc_tol, -1.0, &vo, &st);
        else orc_tensor_solve_bcn(g, &Soln, &Rhs, 1.0, theta * dt, &acoef, ep, c->vlobc, c->vhibc, c->p.visc_tol, -1.0, &vo, &st);
        orc_copy_all(&c->Vsync, &Soln);
        if (c->level > 0) {                       /* :1166-1176: viscflux_reg->FineAdd(tensorflux, ..., dt * dt) */
            orc_fab fl[3]; orc_fab* flp[3];
            for (int d = 0; d < 3; ++d) { fl[d] = orc_alloc(g->n, ORC_FACE[d], 0, 3); flp[d] = &fl[d]; }
            orc_tensor_extensive_flux(g, c->nbox, c->boxes, c->ratio, flp, &Soln, ep, theta, 0, NULL, 2);
            for (int d = 0; d < 3; ++d) { reg_fine_add(c, c->reg_visc, &fl[d], d, 0, Xvel, 3, dt * dt); orc_free(&fl[d]); }
        }
        /* ghost cells outside ext_dir faces back to zero (:987-1008) */
        for (int n = 0; n < 3; ++n) for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
            if (g->periodic[d] || (side == 0 ? c->bc_vel[n].lo[d] : c->bc_vel[n].hi[d]) != ORC_BC_EXT_DIR) continue;
            const int face = side == 0 ? -1 : g->n[d];
            for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i)
                if ((d == 0 ? i : (d == 1 ? j : k)) == face) A4(&c->Vsync, i, j, k, n) = 0.0;
        }
        orc_free(&Rhs); orc_free(&acoef); orc_free(&Soln);
        for (int d = 0; d < 3; ++d) orc_free(&one[d]);
    }
    /* density: not diffusive: Ssync.mult(dt, sigma, 1, ngrow) (:1667-1675) */
    { const size_t N = orc_npts(&c->Ssync); for (size_t q = 0; q < N; ++q) c->Ssync.p[q] *= dt; }
    for (int sn = 1; sn < numscal; ++sn) {
    const int sigma = Density + sn, rho_flag = c->scal_rho_flag[sn];
    const int* slobc = c->slobc + 3 * sn; const int* shibc = c->shibc + 3 * sn;
    if (c->scal_diff[sn] > 0.0) {
        /* Diffusion::diffuse_scalar as the sync solve (NavierStokes.cpp:1590-1640: S_old = {}, S_new = 0, delta_rhs = Ssync, no old-time
         * flux): (alpha - theta dt div D grad) s = dt Ssync, alpha = rho_new for S = rho q (rho_flag 2) else 1; Ssync = s (x rho_new);
         * on a refined level no coarse data are passed upstream: homogeneous coarse/fine data */
        const int cons = rho_flag == 2;
        orc_fab Rhs = orc_alloc(g->n, ORC_CELL, 0, 1), Soln = orc_alloc(g->n, ORC_CELL, 1, 1), acoef = orc_alloc(g->n, ORC_CELL, 0, 1);
        double m = 0.0;
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) {
            A4(&Rhs, i, j, k, 0) = dt * A4(&c->Ssync, i, j, k, sn) * (rho_flag == 1 ? A4(&c->rho_half, i, j, k, 0) : 1.0);   /* Diffusion.cpp:470-475 */
            A4(&acoef, i, j, k, 0) = cons ? A4(Sn, i, j, k, Density) : (rho_flag == 1 ? A4(&c->rho_half, i, j, k, 0) : 1.0);
            if (c->level > 0 && A4(&c->cov, i, j, k, 0) == 0.0) continue;
            if (fabs(A4(&Rhs, i, j, k, 0)) > m) m = fabs(A4(&Rhs, i, j, k, 0));
        }
        orc_abec_level L;
        ns_scalar_level(c, &L, sigma, 1.0, theta * dt, &acoef);
        orc_mg_opts so = c->o; so.maxorder = 2;
        orc_mg_stats st;
        orc_fab cfb = orc_alloc(g->n, ORC_CELL, 1, 3);
        if (c->level > 0) {
            L.nbox = c->nbox; L.boxes = c->boxes;
            for (int d = 0; d < 3; ++d) L.cf_loc[d] = 0.5 * c->ratio * g->dx[d];
            orc_abec_solve_cf(&L, &Soln, &Rhs, slobc, shibc, &cfb, c->p.visc_tol, c->p.visc_tol * m, &so, &st);
            orc_fab fl[3]; orc_fab* flp[3];
            for (int d = 0; d < 3; ++d) { fl[d] = orc_alloc(g->n, ORC_FACE[d], 0, 1); flp[d] = &fl[d]; }
            orc_cf_set_bcval(&cfb, 1, 2);
            orc_abec_extensive_flux(&L, flp, &Soln, theta, 0);
            orc_cf_set_bcval(NULL, 0, 2);
            for (int d = 0; d < 3; ++d) { reg_fine_add(c, c->reg_visc, &fl[d], d, 0, sigma, 1, dt); orc_free(&fl[d]); }   /* NavierStokes.cpp:1630-1638 */
        } else orc_abec_solve(&L, &Soln, &Rhs, slobc, shibc, c->p.visc_tol, c->p.visc_tol * m, &so, &st);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
            A4(&c->Ssync, i, j, k, sn) = A4(&Soln, i, j, k, 0) * (cons ? A4(Sn, i, j, k, Density) : 1.0);
        for (int d = 0; d < 3; ++d) orc_free(&L.b[d]);
        orc_free(&Rhs); orc_free(&Soln); orc_free(&acoef); orc_free(&cfb);
    } else {
        const size_t N = orc_npts(&c->Ssync);
        for (size_t q = 0; q < N; ++q) c->Ssync.p[q + N * sn] *= dt;
    }
    if (c->scal_cons[sn])
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&c->Ssync, i, j, k, sn) += dt * A4(&Delta, i, j, k, sn);
    }
    orc_free(&Delta);
    for (int n = 0; n < numscal; ++n)
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(Sn, i, j, k, 3 + n) += A4(&c->Ssync, i, j, k, n);
    ns_make_rho_curr_time(c);
    if (c->level > 0)
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&c->rho_avg, i, j, k, 0) += A4(&c->Ssync, i, j, k, 0);
    /* interpolate the sync correction to the finer levels (:1697-1725) */
    for (int fl = lev + 1; fl < a->nlev; ++fl) {
        orc_ns_state* f = a->lev[fl];
        orc_fab incr = sync_interp(c, f, &c->Ssync, 0, numscal, c->bc_scal);
        const orc_geom* fg = &f->g;
        orc_fab* Sf = S_NEW(f);
        for (int n = 0; n < numscal; ++n)
        for (int k = 0; k < fg->n[2]; ++k) for (int j = 0; j < fg->n[1]; ++j) for (int i = 0; i < fg->n[0]; ++i)
            if (A4(&f->cov, i, j, k, 0) != 0.0) A4(Sf, i, j, k, 3 + n) += A4(&incr, i, j, k, n);
        ns_make_rho_curr_time(f);
        for (int k = 0; k < fg->n[2]; ++k) for (int j = 0; j < fg->n[1]; ++j) for (int i = 0; i < fg->n[0]; ++i) A4(&f->rho_avg, i, j, k, 0) += A4(&incr, i, j, k, 0);
        orc_free(&incr);
    }
}

/* NavierStokesBase::level_sync (NavierStokesBase.cpp:1927-2044) + Projection::MLsyncProject (Projection.cpp:457-607) */
static void level_sync(orc_amr* a, int lev, int crse_iteration)
{
    orc_ns_state *c = a->lev[lev], *f = a->lev[lev + 1];
    const orc_geom *g = &c->g, *fg = &f->g;
    const double dt = a->dt_level[lev];
    const int crse_dt_ratio = a->n_cycle[lev];
    orc_fill_periodic(&c->Vsync, g, ORC_CELL);
    /* SyncInterp(Vsync -> V_corr), increment = 0 */
    orc_fab Vc = sync_interp(c, f, &c->Vsync, 0, 3, c->bc_vel);
    orc_fab V_corr = orc_alloc(fg->n, ORC_CELL, 1, 3);
    for (int n = 0; n < 3; ++n)
    for (int k = 0; k < fg->n[2]; ++k) for (int j = 0; j < fg->n[1]; ++j) for (int i = 0; i < fg->n[0]; ++i)
        A4(&V_corr, i, j, k, n) = A4(&f->cov, i, j, k, 0) != 0.0 ? A4(&Vc, i, j, k, n) : 0.0;
    orc_free(&Vc);
    /* MLsyncProject */
    orc_fab phi_c = orc_alloc(g->n, ORC_NODE, 1, 1), phi_f = orc_alloc(fg->n, ORC_NODE, 1, 1);
    orc_fab rhnd = orc_alloc(g->n, ORC_NODE, 0, 1);
    syncreg_init_rhs(f, &rhnd);
    {   /* SyncRegister::InitRHS of the literal register (orc_syncreg.c) on the coarse level's boxes */
        orc_ndmf* rb = orc_level_ndmf(c, 0);
        orc_syncreg_init_rhs(f->sync_lit, rb, g, c->p.phys_lo, c->p.phys_hi);
        orc_fab lit = orc_alloc(g->n, ORC_NODE, 0, 1);
        orc_ndmf_to_domain(rb, &lit);
        orc_ndmf_destroy(rb);
        double dmax = 0.0, vmax = 0.0;
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i) {
            if (node_vs_fine(f, i, j, k) == 1) continue;
            if (c->level > 0 && node_class(c, i, j, k) == ND_NONE) continue;
            dmax = fmax(dmax, fabs(A4(&lit, i, j, k, 0) - A4(&rhnd, i, j, k, 0)));
            vmax = fmax(vmax, fabs(A4(&rhnd, i, j, k, 0)));
        }
        orc_syncreg_diff_max = fmax(orc_syncreg_diff_max, vmax > 0.0 ? dmax / vmax : dmax);
        if (orc_syncreg_literal) orc_copy_all(&rhnd, &lit);
        orc_free(&lit);
    }
    {
        int whole = f->nbox == 1;
        for (int d = 0; d < 3 && whole; ++d) if (f->boxes[d] != 0 || f->boxes[3 + d] != fg->n[d] - 1) whole = 0;
        if (whole) orc_setval(&rhnd, 0.0);
    }
    /* scaleVar: sigma = 1/rho (rho_half on the coarse level, rho_avg on the fine one); then average both velocity and sigma down */
    orc_fab sig_c = orc_alloc(g->n, ORC_CELL, 0, 1), sig_f = orc_alloc(fg->n, ORC_CELL, 0, 1);
    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&sig_c, i, j, k, 0) = 1.0 / A4(&c->rho_half, i, j, k, 0);
    for (int k = 0; k < fg->n[2]; ++k) for (int j = 0; j < fg->n[1]; ++j) for (int i = 0; i < fg->n[0]; ++i)
        A4(&sig_f, i, j, k, 0) = A4(&f->cov, i, j, k, 0) != 0.0 ? 1.0 / A4(&f->rho_avg, i, j, k, 0) : 0.0;
    avg_down_cells(f, &V_corr, &c->Vsync, 0, 3);
    avg_down_cells(f, &sig_f, &sig_c, 0, 1);
    orc_fab* vel[2] = {&c->Vsync, &V_corr};
    orc_fab* phi[2] = {&phi_c, &phi_f};
    const orc_fab* sig[2] = {&sig_c, &sig_f};
    /* Projection.cpp:544-569: a sync projection that is not at levels 0-1 changes the level-c velocity; that change enters the sync
     * register of the interface below through the residual of the composite solution on the boundary nodes of level c */
    const int want_resid = lev > 0 && crse_iteration == crse_dt_ratio;
    orc_fab vold_c; vold_c.p = NULL;
    if (want_resid) {
        orc_fill_periodic(&c->Vsync, g, ORC_CELL);
        vold_c = orc_alloc(g->n, ORC_CELL, 1, 3);
        orc_copy_all(&vold_c, &c->Vsync);
    }
    amr_composite_project(a, lev, 2, vel, phi, sig, &rhnd, 1.e-10 /*sync_tol, Projection.cpp:27*/, c->p.proj_abs_tol, 1, 0.0, &a->st_sync);
    if (want_resid) {
        /* SyncRegister::CompAdd (SyncRegister.cpp:321-348): zero under the boxes of level lev+1, then FineAdd with 1/crse_dt_ratio.
         * The residual lives on the boundary of level c, which the next finer level never touches (proper nesting). */
        orc_fab sg1 = orc_alloc(g->n, ORC_CELL, 1, 1);
        for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&sg1, i, j, k, 0) = A4(&sig_c, i, j, k, 0);
        orc_fab r = amr_sync_resid_fine(c, &vold_c, &phi_c, &sg1, NULL);
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
            if (node_vs_fine(f, i, j, k) != 0) A4(&r, i, j, k, 0) = 0.0;
        syncreg_fine_add(c, &r, 1.0 / (double)crse_dt_ratio);
        {   /* literally: crsr_sync_reg->CompAdd(sync_resid_fine, crse_geom, crsr_geom, coarsened boxes of level lev+1, invrat) */
            orc_ndmf* rb = orc_sync_resid_fine_boxes(c, &vold_c, &phi_c, &sg1, NULL);
            int* pb = (int*)malloc(sizeof(int) * 6 * (size_t)f->nbox);
            for (int q = 0; q < f->nbox; ++q) for (int d = 0; d < 3; ++d) { pb[6 * q + d] = f->boxes[6 * q + d] / f->ratio; pb[6 * q + 3 + d] = (f->boxes[6 * q + 3 + d] + 1) / f->ratio - 1; }
            orc_syncreg_comp_add(c->sync_lit, rb, g, &c->crse->g, f->nbox, pb, 1.0 / (double)crse_dt_ratio);
            free(pb); orc_ndmf_destroy(rb);
        }
        orc_free(&r); orc_free(&sg1); orc_free(&vold_c);
    }
    /* add phi to the pressures (with ghost nodes), the projected corrections to the velocities (1 ghost) */
    { const size_t N = orc_npts(P_NEW(c)); for (size_t q = 0; q < N; ++q) P_NEW(c)->p[q] += phi_c.p[q]; }
    { const size_t N = orc_npts(P_NEW(f)); for (size_t q = 0; q < N; ++q) P_NEW(f)->p[q] += phi_f.p[q]; }
    for (int n = 0; n < 3; ++n) {
        for (int k = -1; k <= g->n[2]; ++k) for (int j = -1; j <= g->n[1]; ++j) for (int i = -1; i <= g->n[0]; ++i) A4(S_NEW(c), i, j, k, n) += dt * A4(&c->Vsync, i, j, k, n);
        for (int k = -1; k <= fg->n[2]; ++k) for (int j = -1; j <= fg->n[1]; ++j) for (int i = -1; i <= fg->n[0]; ++i) A4(S_NEW(f), i, j, k, n) += dt * A4(&V_corr, i, j, k, n);
    }
    /* NavierStokesBase.cpp:2018-2040: levels above lev+1 get the interpolated velocity correction (SyncInterp, increment = 1, x dt) and
     * pressure correction (SyncProjInterp: node_bilinear_interp of phi, added to P_new AND P_old), then computeGradP at both times */
    for (int l2 = lev + 2; l2 < a->nlev; ++l2) {
        orc_ns_state* ff = a->lev[l2];
        const orc_geom* gg = &ff->g;
        orc_fab Vi = sync_interp(f, ff, &V_corr, 0, 3, f->bc_vel);
        int ratio = 1;
        for (const orc_ns_state* q = ff; q != f; q = q->crse) ratio *= q->ratio;
        for (int n = 0; n < 3; ++n)
        for (int k = 0; k < gg->n[2]; ++k) for (int j = 0; j < gg->n[1]; ++j) for (int i = 0; i < gg->n[0]; ++i)
            if (A4(&ff->cov, i, j, k, 0) != 0.0) A4(S_NEW(ff), i, j, k, n) += dt * A4(&Vi, i, j, k, n);
        orc_free(&Vi);
        orc_nodal_fill_bc(fg, &phi_f, f->nlobc, f->nhibc);
        for (int k = 0; k <= gg->n[2]; ++k) for (int j = 0; j <= gg->n[1]; ++j) for (int i = 0; i <= gg->n[0]; ++i) {
            if (node_class(ff, i, j, k) == ND_NONE) continue;
            const int fi[3] = {i, j, k};
            int c0[3]; double w[3];
            for (int d = 0; d < 3; ++d) { c0[d] = fi[d] / ratio; w[d] = (double)(fi[d] - c0[d] * ratio) / (double)ratio; }
            double v = 0.0;
            for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
                const double ww = (cx ? w[0] : 1.0 - w[0]) * (cy ? w[1] : 1.0 - w[1]) * (cz ? w[2] : 1.0 - w[2]);
                if (ww != 0.0) v += ww * A4(&phi_f, c0[0] + cx, c0[1] + cy, c0[2] + cz, 0);
            }
            A4(P_NEW(ff), i, j, k, 0) += v; A4(P_OLD(ff), i, j, k, 0) += v;
        }
        for (int which = 0; which < 2; ++which) {            /* computeGradP(prevTime), computeGradP(curTime) */
            orc_fab* G = which ? GP_NEW(ff) : GP_OLD(ff);
            orc_fab gp = orc_alloc(gg->n, ORC_CELL, 0, 3);
            orc_nodal_compgrad(gg, &gp, which ? P_NEW(ff) : P_OLD(ff));
            for (int n = 0; n < 3; ++n)
            for (int k = 0; k < gg->n[2]; ++k) for (int j = 0; j < gg->n[1]; ++j) for (int i = 0; i < gg->n[0]; ++i)
                if (A4(&ff->cov, i, j, k, 0) != 0.0) A4(G, i, j, k, n) = A4(&gp, i, j, k, n);
            orc_free(&gp);
            ns_fill_gp(ff, G, which ? 0.5 * (ff->pt_new[0] + ff->pt_new[1]) : 0.5 * (ff->pt_old[0] + ff->pt_old[1]));
        }
    }
    orc_free(&V_corr); orc_free(&phi_c); orc_free(&phi_f); orc_free(&rhnd); orc_free(&sig_c); orc_free(&sig_f);
}

/* NavierStokesBase::post_timestep (NavierStokesBase.cpp:2546-2636) */
static void post_timestep(orc_amr* a, int lev, int crse_iteration)
{
    orc_ns_state* s = a->lev[lev];
    if (lev < a->nlev - 1) {
        const char* dbg = getenv("ORC_AMR_DEBUG");       /* debugging aid: bit 0 skip reflux, 1 skip mac_sync, 2 skip level_sync */
        const int skip = dbg ? atoi(dbg) : 0;
        if (!(skip & 1)) reflux(a, lev);
        avg_down(a, lev);
        if (!(skip & 2)) mac_sync(a, lev);
        if (!(skip & 4)) level_sync(a, lev, crse_iteration);
    }
    if (lev > 0) {                       /* incrPAvg */
        const double alpha = 1.0 / (double)a->n_cycle[lev];
        const orc_geom* g = &s->g;
        for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
            A4(&s->p_avg, i, j, k, 0) += alpha * A4(P_NEW(s), i, j, k, 0);
    }
}

/* Amr::timeStep */
void orc_amr_regrid_from(orc_amr* a, int lbase, double cur_time, int nnew, const int* nbox, const int* boxes);
/* regrids to replay during the next coarse step (the grids come from the product, see orc_amr_regrid): applied at the start of the step
 * of level `lev` at `time` for every scheduled base level >= lev, as Amr::timeStep does */
typedef struct { int lbase, nnew, done; double time; int nbox[8]; int* boxes; } regrid_event;
static regrid_event g_events[64];
static int g_nevents = 0;
void orc_amr_schedule_regrid(int lbase, double time, int nnew, const int* nbox, const int* boxes)
{
    regrid_event* e = &g_events[g_nevents++];
    e->lbase = lbase; e->nnew = nnew; e->done = 0; e->time = time;
    int tot = 0;
    for (int q = 0; q < nnew; ++q) { e->nbox[q] = nbox[q]; tot += nbox[q]; }
    e->boxes = (int*)malloc(sizeof(int) * 6 * (size_t)(tot > 0 ? tot : 1));
    memcpy(e->boxes, boxes, sizeof(int) * 6 * (size_t)tot);
}
void orc_amr_clear_regrid_schedule(void) { for (int q = 0; q < g_nevents; ++q) free(g_events[q].boxes); g_nevents = 0; }

static void time_step(orc_amr* a, int lev, double time, int iteration, int niter)
{
    for (int q = 0; q < g_nevents; ++q) {
        regrid_event* e = &g_events[q];
        if (e->done || e->lbase < lev || e->lbase >= a->nlev) continue;
        if (fabs(e->time - time) > 1.e-9 * fmax(1.0, fabs(time)) + 1.e-300) continue;
        orc_amr_regrid_from(a, e->lbase, time, e->nnew, e->nbox, e->boxes);
        e->done = 1;
    }
    orc_ns_state* s = a->lev[lev];
    s->time = time;
    const double dt_new = ns_advance(s, a->dt_level[lev], iteration, niter);
    a->dt_min[lev] = iteration == 1 ? dt_new : fmin(a->dt_min[lev], dt_new);
    s->time = time + a->dt_level[lev];
    s->nstep += 1;
    if (lev < a->nlev - 1) {
        const int nc = a->n_cycle[lev + 1];
        for (int i = 1; i <= nc; ++i) time_step(a, lev + 1, time + (i - 1) * a->dt_level[lev + 1], i, nc);
    }
    post_timestep(a, lev, iteration);
}

/* ------------------------------------------------------------------------------------------------------------------------ */
orc_amr* orc_amr_create(const orc_geom* g0, const orc_ns_params* p, const orc_mg_opts* o, int nlev, int ratio, const int* nbox, const int* boxes)
{
    if (ratio != 2 || nlev < 1 || nlev > 8) { fprintf(stderr, "orc_amr_create: ratio 2, 1..8 levels\n"); return NULL; }
    orc_amr* a = (orc_amr*)calloc(1, sizeof(orc_amr));
    a->nlev = nlev;
    a->stop_time = -1.0;
    orc_geom g = *g0;
    const int* bp = boxes;
    for (int l = 0; l < nlev; ++l) {
        if (l > 0) for (int d = 0; d < 3; ++d) { g.n[d] *= ratio; g.dx[d] /= (double)ratio; }
        orc_ns_state* s = orc_ns_create(&g, p, o);
        if (!s) { free(a); return NULL; }
        a->lev[l] = s;
        s->level = l; s->ratio = l > 0 ? ratio : 1;
        a->n_cycle[l] = l > 0 ? ratio : 1;
        if (l > 0) {
            s->crse = a->lev[l - 1]; a->lev[l - 1]->fine = s;
            s->nbox = nbox[l];
            s->boxes = (int*)malloc(sizeof(int) * 6 * (size_t)s->nbox);
            memcpy(s->boxes, bp, sizeof(int) * 6 * (size_t)s->nbox);
            bp += 6 * s->nbox;
            s->cov = orc_alloc(g.n, ORC_CELL, 0, 1);
            for (int b = 0; b < s->nbox; ++b) {
                const int* bx = s->boxes + 6 * b;
                for (int k = bx[2]; k <= bx[5]; ++k) for (int j = bx[1]; j <= bx[4]; ++j) for (int i = bx[0]; i <= bx[3]; ++i) A4(&s->cov, i, j, k, 0) = 1.0;
            }
            s->rho_avg = orc_alloc(g.n, ORC_CELL, 1, 1); s->p_avg = orc_alloc(g.n, ORC_NODE, 0, 1);
            const orc_geom* cg = &s->crse->g;
            for (int d = 0; d < 3; ++d) {
                s->reg_adv[d] = orc_alloc(cg->n, ORC_FACE[d], 0, s->nstate);
                s->reg_visc[d] = orc_alloc(cg->n, ORC_FACE[d], 0, s->nstate);
                s->reg_mac[d] = orc_alloc(cg->n, ORC_FACE[d], 0, 1);
            }
            s->sync_reg = orc_alloc(cg->n, ORC_NODE, 0, 1);
            s->sync_lit = orc_syncreg_create(s->nbox, s->boxes, s->ratio);
            s->crse->Vsync = orc_alloc(cg->n, ORC_CELL, 1, 3);
            s->crse->Ssync = orc_alloc(cg->n, ORC_CELL, 1, s->nstate - 3);
        }
    }
    return a;
}
void orc_amr_destroy(orc_amr* a)
{
    for (int l = 0; l < a->nlev; ++l) orc_ns_destroy(a->lev[l]);
    free(a);
}
orc_ns_state* orc_amr_level(orc_amr* a, int lev) { return a->lev[lev]; }
const orc_fab* orc_amr_cov(orc_amr* a, int lev) { return a->lev[lev]->cov.p ? &a->lev[lev]->cov : NULL; }
double orc_amr_time(const orc_amr* a) { return a->lev[0]->time; }
double orc_amr_dt(const orc_amr* a, int lev) { return a->dt_level[lev]; }
void orc_amr_sync_stats(const orc_amr* a, orc_mg_stats* st) { *st = a->st_sync; }

/* NavierStokes::post_init (NavierStokes.cpp:1254-1299) for the whole hierarchy.  The caller has filled S_new of every level (the
 * cells of the level; everything else is ignored). */
void orc_amr_post_init(orc_amr* a, double stop_time)
{
    const int nl = a->nlev, fin = nl - 1;
    a->stop_time = stop_time;
    for (int l = 0; l < nl; ++l) {
        orc_ns_state* s = a->lev[l];
        orc_setval(P_NEW(s), 0.0); orc_setval(P_OLD(s), 0.0); orc_setval(GP_NEW(s), 0.0); orc_setval(GP_OLD(s), 0.0);
        s->time = 0.0; s->nstep = 0;
        ns_set_time_level(s, 0.0, 0.0, 0.0);
    }
    orc_fab* vel[8]; orc_fab* phi[8]; const orc_fab* sigp[8]; orc_fab sig[8], vv[8];
    orc_fab rc[8]; orc_fab* rcp[8];
    const int have_divu = a->lev[0]->have_divu;
    for (int l = 0; l < nl; ++l) { rc[l].p = NULL; rcp[l] = NULL; }
    if (have_divu)                          /* NavierStokes::initData (NavierStokes.cpp:457-479), level by level: rho at both times, divu, dsdt = 0 */
        for (int l = 0; l < nl; ++l) {
            orc_ns_state* s = a->lev[l];
            const orc_geom* g = &s->g;
            ns_make_rho_curr_time(s);
            orc_copy_all(&s->rho_ptime, &s->rho_ctime);
            ns_calc_divu(s, 1);
            for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(S_NEW(s), i, j, k, s->Dsdt) = 0.0;
        }
    /* ---- post_init_state (NavierStokesBase.cpp:2369-2439) ---- */
    const orc_ns_params* p = &a->lev[0]->p;
    if (p->init_vel_iter <= 0) {
        for (int l = 0; l < nl; ++l) { orc_setval(P_OLD(a->lev[l]), 0.0); orc_setval(GP_OLD(a->lev[l]), 0.0); }
    } else
    for (int iter = 0; iter < p->init_vel_iter; ++iter) {           /* Projection::initialVelocityProject */
        for (int l = 0; l < nl; ++l) {
            orc_ns_state* s = a->lev[l];
            orc_setval(P_OLD(s), 0.0);
            sig[l] = orc_alloc(s->g.n, ORC_CELL, 0, 1); orc_setval(&sig[l], 1.0);       /* rho_wgt_vel_proj = 0 */
            vv[l] = *S_NEW(s); vv[l].nc = 3;
            vel[l] = &vv[l]; phi[l] = P_OLD(s); sigp[l] = &sig[l];
            if (have_divu) {                /* rhcc = -getDivCond(cur_divu_time), Projection.cpp:732-743, 783-788 */
                const orc_geom* g = &s->g;
                rc[l] = orc_alloc(g->n, ORC_CELL, 0, 1); rcp[l] = &rc[l];
                for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(&rc[l], i, j, k, 0) = -A4(S_NEW(s), i, j, k, s->Divu);
            }
        }
        amr_composite_project_rhcc(a, 0, nl, vel, phi, sigp, NULL, have_divu ? rcp : NULL, p->proj_tol, p->proj_abs_tol, 0, 1.0, &a->lev[0]->st_nodal);
        for (int l = 0; l < nl; ++l) if (rc[l].p) { orc_free(&rc[l]); rc[l].p = NULL; }
        for (int l = 0; l < nl; ++l) {
            orc_ns_state* s = a->lev[l];
            orc_setval(P_OLD(s), 0.0); orc_setval(P_NEW(s), 0.0); orc_setval(GP_OLD(s), 0.0); orc_setval(GP_NEW(s), 0.0);
            orc_free(&sig[l]);
        }
    }
    for (int l = 0; l < nl; ++l) a->lev[l]->initial_step = 1;
    for (int l = fin - 1; l >= 0; --l) avg_down(a, l);
    if (fabs(p->gravity) > 0.0) {                                    /* Projection::initialPressureProject (Projection.cpp:841-960) */
        for (int l = 0; l < nl; ++l) {
            orc_ns_state* s = a->lev[l];
            const orc_geom* g = &s->g;
            sig[l] = orc_alloc(g->n, ORC_CELL, 0, 1);
            for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
                A4(&sig[l], i, j, k, 0) = ns_covered(s, i, j, k) ? 1.0 / A4(S_NEW(s), i, j, k, Density) : 0.0;
            vv[l] = orc_alloc(g->n, ORC_CELL, 1, 3);
            { const size_t N = orc_npts(&vv[l]); for (size_t q = 0; q < N; ++q) vv[l].p[q + 2 * N] = p->gravity; }
            vel[l] = &vv[l]; phi[l] = P_NEW(s); sigp[l] = &sig[l];
        }
        /* set_outflow_bcs(INITIAL_PRESS, c_lev = 0 .. f_lev; Projection.cpp:893-905, 1776-1803): the finest level that covers the whole
         * strip of an outflow face computes the hydrostatic data, putDown (:1656-1712) injects them into the coarser levels.  With one
         * outflow face or none this is: finest covering level first, injection below it */
        {
            int done = 0;
            for (int l = nl - 1; l >= 0; --l) {
                orc_ns_state* s = a->lev[l];
                const orc_geom* g = &s->g;
                if (!done) {
                    orc_fab rho = ns_fillpatch_time(s, s->st_new, 0, Density, 1, 1);
                    orc_fab mark = orc_alloc(g->n, ORC_NODE, 1, 1);
                    orc_setval(&mark, -7.e33);
                    ns_set_outflow_bcs(s, &mark, &rho);
                    for (int k = 0; k <= g->n[2]; ++k) for (int j = 0; j <= g->n[1]; ++j) for (int i = 0; i <= g->n[0]; ++i)
                        if (A4(&mark, i, j, k, 0) != -7.e33) { A4(phi[l], i, j, k, 0) = A4(&mark, i, j, k, 0); done = 1; }
                    orc_free(&rho); orc_free(&mark);
                } else {
                    const orc_ns_state* f = a->lev[l + 1];
                    const int r = f->ratio;
                    for (int D = 0; D < 2; ++D) for (int side = 0; side < 2; ++side) {
                        if (g->periodic[D] || (side == 0 ? s->p.phys_lo[D] : s->p.phys_hi[D]) != 2 /*Outflow*/) continue;
                        for (int k = 0; k <= g->n[2]; ++k) for (int t = 0; t <= g->n[1 - D]; ++t) {
                            int q[3]; q[D] = side == 0 ? 0 : g->n[D]; q[1 - D] = t; q[2] = k;
                            A4(phi[l], q[0], q[1], q[2], 0) = A4(phi[l + 1], q[0] * r, q[1] * r, q[2] * r, 0);
                        }
                    }
                }
            }
        }
        amr_composite_project(a, 0, nl, vel, phi, sigp, NULL, p->proj_tol, p->proj_abs_tol, 0, 0.0, &a->lev[0]->st_nodal);
        for (int l = 0; l < nl; ++l) {
            orc_ns_state* s = a->lev[l];
            orc_copy_all(P_OLD(s), P_NEW(s)); orc_copy_all(GP_OLD(s), GP_NEW(s));
            orc_free(&sig[l]); orc_free(&vv[l]);
        }
    }
    /* ---- post_init_estDT (NavierStokesBase.cpp:2307-2362) ---- */
    double dt_save[8], dt_init = 1.0e+100;
    int nc_save[8];
    for (int k = 0; k < nl; ++k) {
        nc_save[k] = a->n_cycle[k];
        dt_save[k] = p->init_shrink * ns_est_time_step(a->lev[k]);          /* initialTimeStep */
        int n_factor = 1;
        for (int m = fin; m > k; --m) n_factor *= a->n_cycle[m];
        dt_init = fmin(dt_init, dt_save[k] / (double)n_factor);
    }
    double dt0 = dt_save[0];
    { int n_factor = 1; for (int k = 0; k < nl; ++k) { n_factor *= nc_save[k]; dt0 = fmin(dt0, n_factor * dt_save[k]); } }
    if (stop_time >= 0.0) { const double eps = 0.0001 * dt0; if (0.0 + dt0 > stop_time - eps) dt0 = stop_time - 0.0; }
    { int n_factor = 1; for (int k = 0; k < nl; ++k) { n_factor *= nc_save[k]; dt_save[k] = dt0 / (double)n_factor; } }
    for (int k = 0; k < nl; ++k) { a->dt_level[k] = dt_init; a->n_cycle[k] = 1; ns_set_time_level(a->lev[k], 0.0, dt_init, dt_init); }
    /* ---- post_init_press (NavierStokes.cpp:1306-1432) ---- */
    if (p->init_iter > 0) {
        for (int l = 0; l < nl; ++l) a->lev[l]->initial_iter = 1;
        for (int iter = 0; iter < p->init_iter; ++iter) {
            for (int k = 0; k < nl; ++k) ns_advance(a->lev[k], dt_init, 1, 1);
            /* Projection::initialSyncProject (Projection.cpp:970-1185) */
            for (int l = 0; l < nl; ++l) {
                orc_ns_state* s = a->lev[l];
                const orc_geom* g = &s->g;
                orc_setval(P_OLD(s), 0.0);
                orc_fab *Un = S_NEW(s), *Uo = S_OLD(s);
                for (int n = 0; n < 3; ++n)
                for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
                    A4(Un, i, j, k, n) = (A4(Un, i, j, k, n) - A4(Uo, i, j, k, n)) * (1. / dt_init);      /* ConvertUnew */
                sig[l] = orc_alloc(g->n, ORC_CELL, 0, 1);
                for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
                    A4(&sig[l], i, j, k, 0) = ns_covered(s, i, j, k) ? 1.0 / A4(&s->rho_half, i, j, k, 0) : 0.0;
                vv[l] = *Un; vv[l].nc = 3;
                vel[l] = &vv[l]; phi[l] = P_OLD(s); sigp[l] = &sig[l];
                if (have_divu) {            /* rhcc = -(divu(strt_time + dt) - divu(strt_time)) / dt, Projection.cpp:1008-1075, 1142-1148 */
                    rc[l] = orc_alloc(g->n, ORC_CELL, 0, 1); rcp[l] = &rc[l];
                    for (int k = 0; k < g->n[2]; ++k) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i)
                        A4(&rc[l], i, j, k, 0) = -((A4(S_NEW(s), i, j, k, s->Divu) - A4(S_OLD(s), i, j, k, s->Divu)) * (1. / dt_init));
                }
            }
            for (int l = fin; l >= 1; --l) avg_down_cells(a->lev[l], vel[l], vel[l - 1], 0, 3);
            amr_composite_project_rhcc(a, 0, nl, vel, phi, sigp, NULL, have_divu ? rcp : NULL, p->proj_tol, p->proj_abs_tol, 1, 0.0, &a->lev[0]->st_nodal);
            for (int l = 0; l < nl; ++l) if (rc[l].p) { orc_free(&rc[l]); rc[l].p = NULL; }
            for (int l = 0; l < nl; ++l) {
                orc_ns_state* s = a->lev[l];
                const size_t N = orc_npts(P_NEW(s));
                for (size_t q = 0; q < N; ++q) P_NEW(s)->p[q] += P_OLD(s)->p[q];
                orc_free(&sig[l]);
            }
            for (int k = fin - 1; k >= 0; --k) avg_down(a, k);
            for (int k = 0; k < nl; ++k) {                           /* resetState(strt_time, dt_init, dt_init) */
                orc_ns_state* s = a->lev[k];
                s->inew = 1 - s->inew;
                if (s->have_divu) {         /* Dsdt_Type is not reset (NavierStokesBase.cpp:2669-2676) */
                    const orc_geom* g = &s->g;
                    for (int kk = 0; kk < g->n[2]; ++kk) for (int j = 0; j < g->n[1]; ++j) for (int i = 0; i < g->n[0]; ++i) A4(S_NEW(s), i, j, kk, s->Dsdt) = A4(S_OLD(s), i, j, kk, s->Dsdt);
                }
                orc_copy_all(P_OLD(s), P_NEW(s)); orc_copy_all(GP_OLD(s), GP_NEW(s));
                ns_set_time_level(s, 0.0, dt_init, dt_init);
                s->initial_iter = 0;
            }
        }
    }
    for (int k = 0; k < nl; ++k) {
        orc_ns_state* s = a->lev[k];
        s->initial_step = 0; s->initial_iter = 0;
        ns_set_time_level(s, 0.0, dt_save[k], dt_save[k]);
        a->dt_level[k] = dt_save[k]; a->n_cycle[k] = nc_save[k]; a->dt_min[k] = 1.e200;
        s->dt = dt_save[k];
    }
    a->level_steps = 0;
}

/* Amr::coarseTimeStep: computeNewDt (NavierStokesBase.cpp:945-1035) + timeStep(0) */
/* NavierStokesBase::computeNewDt (NavierStokesBase.cpp:945-1035); post_regrid: limited by the pre-regrid dt (:973-982) */
static void compute_new_dt(orc_amr* a, int post_regrid)
{
    const int nl = a->nlev;
    const orc_ns_params* p = &a->lev[0]->p;
    const double cur_time = a->lev[0]->time;
    for (int i = 0; i < nl; ++i) a->dt_min[i] = fmin(a->dt_min[i], ns_est_time_step(a->lev[i]));
    if (p->fixed_dt <= 0.0) for (int i = 0; i < nl; ++i) a->dt_min[i] = fmin(a->dt_min[i], post_regrid ? a->dt_level[i] : p->change_max * a->dt_level[i]);
    double dt_0 = 1.0e+100;
    int n_factor = 1;
    for (int i = 0; i < nl; ++i) { n_factor *= a->n_cycle[i]; dt_0 = fmin(dt_0, n_factor * a->dt_min[i]); }
    const double eps = 0.0001 * dt_0;
    if (a->stop_time >= 0.0 && cur_time + dt_0 > a->stop_time - eps) dt_0 = a->stop_time - cur_time;
    n_factor = 1;
    for (int i = 0; i < nl; ++i) { n_factor *= a->n_cycle[i]; a->dt_level[i] = dt_0 / (double)n_factor; }
}

/* Amr::regrid from level 0 with GIVEN new grids (the grid generation itself -- tags, clustering -- is tested on its own,
 * orc_regrid.c): the data of the new levels as NavierStokesBase::init(AmrLevel& old) / init() fill them
 * (NavierStokesBase.cpp:1713-1806): FillPatch of State / Gradp (the old level's cells where it existed, cell_cons_interp of the
 * rebuilt coarser level elsewhere) and of Press (node_bilinear_interp + the old level's nodes); time levels
 * setTimeLevel(cur_time, dt_old, dt_new); a new level starts with dt = dt_crse / ratio.  Call between coarse steps; the caller then
 * uses orc_amr_coarse_step_post_regrid for the next step (computeNewDt with post_regrid_flag = 1). */
void orc_amr_regrid_from(orc_amr* a, int lbase, double cur_time, int nnew, const int* nbox, const int* boxes);
void orc_amr_regrid(orc_amr* a, int nfine, const int* nbox, const int* boxes) { orc_amr_regrid_from(a, 0, a->lev[0]->time, nfine, nbox, boxes); }

/* Amr::regrid(lbase, time): the levels up to lbase keep their grids, nnew levels above it are (re)built from the given boxes (nbox[q],
 * boxes: level lbase + 1 + q); cur_time: the time all of them have reached (inside a coarse step for lbase > 0) */
void orc_amr_regrid_from(orc_amr* a, int lbase, double cur_time, int nnew, const int* nbox, const int* boxes)
{
    const int old_nlev = a->nlev, ratio = 2, nfine = lbase + nnew;
    orc_ns_state* old[8];
    for (int l = 0; l < 8; ++l) old[l] = l < old_nlev ? a->lev[l] : NULL;
    const int* bp = boxes;
    a->lev[lbase]->fine = NULL;
    for (int l = lbase + 1; l <= nfine; ++l) {
        orc_ns_state* c = a->lev[l - 1];
        orc_geom g = c->g;
        for (int d = 0; d < 3; ++d) { g.n[d] *= ratio; g.dx[d] /= (double)ratio; }
        orc_ns_state* s = orc_ns_create(&g, &c->p, &c->o);
        orc_ns_state* ol = old[l];
        s->level = l; s->ratio = ratio;
        s->crse = c; c->fine = s;
        s->nbox = nbox[l - lbase - 1];
        s->boxes = (int*)malloc(sizeof(int) * 6 * (size_t)s->nbox);
        memcpy(s->boxes, bp, sizeof(int) * 6 * (size_t)s->nbox);
        bp += 6 * s->nbox;
        s->cov = orc_alloc(g.n, ORC_CELL, 0, 1);
        for (int b = 0; b < s->nbox; ++b) {
            const int* bx = s->boxes + 6 * b;
            for (int k = bx[2]; k <= bx[5]; ++k) for (int j = bx[1]; j <= bx[4]; ++j) for (int i = bx[0]; i <= bx[3]; ++i) A4(&s->cov, i, j, k, 0) = 1.0;
        }
        s->rho_avg = orc_alloc(g.n, ORC_CELL, 1, 1); s->p_avg = orc_alloc(g.n, ORC_NODE, 0, 1);
        const orc_geom* cg = &c->g;
        for (int d = 0; d < 3; ++d) {
            s->reg_adv[d] = orc_alloc(cg->n, ORC_FACE[d], 0, s->nstate);
            s->reg_visc[d] = orc_alloc(cg->n, ORC_FACE[d], 0, s->nstate);
            s->reg_mac[d] = orc_alloc(cg->n, ORC_FACE[d], 0, 1);
        }
        s->sync_reg = orc_alloc(cg->n, ORC_NODE, 0, 1);
        s->sync_lit = orc_syncreg_create(s->nbox, s->boxes, s->ratio);
        if (!c->Vsync.p) { c->Vsync = orc_alloc(cg->n, ORC_CELL, 1, 3); c->Ssync = orc_alloc(cg->n, ORC_CELL, 1, c->nstate - 3); }
        /* times */
        const double dt_new = ol ? a->dt_level[l] : a->dt_level[l - 1] / (double)ratio;
        const double dt_old = ol ? ol->st_new - ol->st_old : (c->st_new - c->st_old) / (double)ratio;
        a->dt_level[l] = dt_new; a->n_cycle[l] = ratio;
        if (!ol) a->dt_min[l] = 1.e200;
        s->time = cur_time; s->nstep = ol ? ol->nstep : 0; s->dt = dt_new;
        ns_set_time_level(s, cur_time, dt_old, dt_new);
        s->initial_step = 0; s->initial_iter = 0;
        /* data.  A temporary view of the new level that still holds the OLD level's coverage and data serves as FillPatch source:
         * whole-domain arrays interpolated from the (new) coarser level, overwritten by the old level's cells where it existed */
        {
            orc_ns_state src = *s;                   /* shallow: geometry, BCs, crse pointer of the NEW level */
            if (ol) { src.cov = ol->cov; src.nbox = ol->nbox; src.boxes = ol->boxes; src.S[0] = ol->S[0]; src.S[1] = ol->S[1]; src.inew = ol->inew;
                      src.Gp[0] = ol->Gp[0]; src.Gp[1] = ol->Gp[1]; src.P[0] = ol->P[0]; src.P[1] = ol->P[1]; src.pnew = ol->pnew;
                      src.st_new = cur_time; src.st_old = ol->st_old; src.pt_new[0] = ol->pt_new[0]; src.pt_new[1] = ol->pt_new[1];
                      src.pt_old[0] = ol->pt_old[0]; src.pt_old[1] = ol->pt_old[1]; }
            else { orc_fab none = orc_alloc(g.n, ORC_CELL, 0, 1); src.cov = none; src.nbox = 0; src.st_new = cur_time; src.st_old = cur_time - dt_old; }
            src.crse = c;
            orc_fab Sv = ns_fillpatch_time(&src, cur_time, 0, Xvel, 3, 1);
            orc_fab Sq[ORC_MAXSLOT];                  /* the scalars, and divu / dsdt (NavierStokesBase.cpp:1742-1754, 1800-1805) */
            for (int n = 0; n < s->nalloc - 3; ++n) Sq[n] = ns_fillpatch_time(&src, cur_time, 0, Density + n, 1, 1);
            const double tp = 0.5 * (s->pt_new[0] + s->pt_new[1]);
            orc_fab Gv = ns_fillpatch_time(&src, ol ? 0.5 * (ol->pt_new[0] + ol->pt_new[1]) : tp, 1, 0, 3, 1);
            for (int q = 0; q < 2; ++q) {
                for (int n = 0; n < 3; ++n)
                for (int k = -1; k <= g.n[2]; ++k) for (int j = -1; j <= g.n[1]; ++j) for (int i = -1; i <= g.n[0]; ++i) {
                    A4(&s->S[q], i, j, k, n) = A4(&Sv, i, j, k, n);
                    A4(&s->Gp[q], i, j, k, n) = A4(&Gv, i, j, k, n);
                }
                for (int n = 0; n < s->nalloc - 3; ++n)
                for (int k = -1; k <= g.n[2]; ++k) for (int j = -1; j <= g.n[1]; ++j) for (int i = -1; i <= g.n[0]; ++i)
                    A4(&s->S[q], i, j, k, Density + n) = A4(&Sq[n], i, j, k, 0);
            }
            orc_free(&Sv); for (int n = 0; n < s->nalloc - 3; ++n) orc_free(&Sq[n]); orc_free(&Gv);
            if (!ol) orc_free(&src.cov);
            /* pressure: node_bilinear_interp of the coarse pressure on every node of the new level, then the old level's nodes */
            const orc_fab* Pc = P_NEW(c);
            for (int k = 0; k <= g.n[2]; ++k) for (int j = 0; j <= g.n[1]; ++j) for (int i = 0; i <= g.n[0]; ++i) {
                if (node_class(s, i, j, k) == ND_NONE) continue;
                const int fi[3] = {i, j, k};
                int c0[3]; double w[3];
                for (int d = 0; d < 3; ++d) { c0[d] = fi[d] / ratio; w[d] = (double)(fi[d] - c0[d] * ratio) / (double)ratio; }
                double v = 0.0;
                for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
                    const double ww = (cx ? w[0] : 1.0 - w[0]) * (cy ? w[1] : 1.0 - w[1]) * (cz ? w[2] : 1.0 - w[2]);
                    if (ww != 0.0) v += ww * A4(Pc, c0[0] + cx, c0[1] + cy, c0[2] + cz, 0);
                }
                if (ol && node_class(ol, i, j, k) != ND_NONE) v = A4(P_NEW(ol), i, j, k, 0);
                A4(&s->P[0], i, j, k, 0) = v; A4(&s->P[1], i, j, k, 0) = v;
            }
        }
        ns_make_rho_curr_time(s);
        a->lev[l] = s;
    }
    for (int l = lbase + 1; l < old_nlev; ++l) if (old[l]) orc_ns_destroy(old[l]);
    for (int l = nfine + 1; l < 8; ++l) a->lev[l] = NULL;
    a->nlev = nfine + 1;
}

/* the coarse step that follows a regrid: computeNewDt as usual (orc_amr_compute_new_dt, before the regrid), then -- only with
 * amr.compute_new_dt_on_regrid = 1, Amr::timeStep; the default is 0 -- again with post_regrid_flag = 1 */
double orc_amr_coarse_step_post_regrid(orc_amr* a, int compute_new_dt_on_regrid)
{
    if (compute_new_dt_on_regrid) compute_new_dt(a, 1);
    time_step(a, 0, a->lev[0]->time, 1, 1);
    a->level_steps += 1;
    for (int i = 0; i < a->nlev; ++i) a->lev[i]->dt = a->dt_level[i];
    return a->dt_level[0];
}
void orc_amr_compute_new_dt(orc_amr* a) { if (a->level_steps > 0) compute_new_dt(a, 0); }

double orc_amr_coarse_step(orc_amr* a)
{
    const int nl = a->nlev;
    const orc_ns_params* p = &a->lev[0]->p;
    const double cur_time = a->lev[0]->time;
    if (a->level_steps > 0) {
        for (int i = 0; i < nl; ++i) a->dt_min[i] = fmin(a->dt_min[i], ns_est_time_step(a->lev[i]));
        if (p->fixed_dt <= 0.0) for (int i = 0; i < nl; ++i) a->dt_min[i] = fmin(a->dt_min[i], p->change_max * a->dt_level[i]);
        double dt_0 = 1.0e+100;
        int n_factor = 1;
        for (int i = 0; i < nl; ++i) { n_factor *= a->n_cycle[i]; dt_0 = fmin(dt_0, n_factor * a->dt_min[i]); }
        const double eps = 0.0001 * dt_0;
        if (a->stop_time >= 0.0 && cur_time + dt_0 > a->stop_time - eps) dt_0 = a->stop_time - cur_time;
        n_factor = 1;
        for (int i = 0; i < nl; ++i) { n_factor *= a->n_cycle[i]; a->dt_level[i] = dt_0 / (double)n_factor; }
    }
    time_step(a, 0, cur_time, 1, 1);
    a->level_steps += 1;
    for (int i = 0; i < nl; ++i) a->lev[i]->dt = a->dt_level[i];
    return a->dt_level[0];
}
