// iamr_amd/csrc/amrregrid.hip -- regridding of the hierarchy (SURVEY f1 on top of a18): Amr::regrid from level 0 as IAMR drives it.
//
// Reference:
//   NavierStokes::errorEst / error_setup     Source/NS_error.cpp:10-145 (amr.refinement_indicators: value_greater, value_less,
//                                            vorticity_greater, adjacent_difference_greater, in_box_lo/hi, max_level)
//   NavierStokesBase::init(AmrLevel& old)    Source/NavierStokesBase.cpp:1713-1754 (a level that existed: FillPatch of State, Press, Gradp)
//   NavierStokesBase::init()                 Source/NavierStokesBase.cpp:1759-1806 (a new level: FillCoarsePatch, dt = dt_crse / ratio)
//   NavierStokesBase::computeNewDt           Source/NavierStokesBase.cpp:971-982 (post_regrid_flag = 1)
//   Amr::regrid / AmrMesh::MakeNewGrids      upstream AMReX (tag, buffer, cluster on the blocking-factor lattice, proper nesting)
//
// Grid generation, top-down: for lev = finest-possible .. 0 the tags of level lev are the error tags of its data plus the cells under
// the already generated level lev+2 grids grown by the nesting buffer, so that every new level nests properly in the one below
// (3 ghost cells of the Godunov stencil + 1 coarse interpolation cell, the condition AmrNS's constructor checks).
#include "operators.h"
#include "launch.h"
#include "amrns.h"
#include <cstring>
#include <algorithm>
#include <cmath>

namespace iamrx {

// regrid.hip

namespace {

// DistributionMapping of a new level: the boxes, largest first, each to the rank with the least cells so far (knapsack; ties -> lowest
// rank).  A function of the box list alone: every rank computes the same owners.
std::vector<int> distribute_boxes(const std::vector<BoxD>& boxes, int nranks)
{
    std::vector<int> owner(boxes.size(), 0);
    if (nranks <= 1) return owner;
    std::vector<size_t> order(boxes.size());
    for (size_t q = 0; q < order.size(); ++q) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return boxes[a].npts() > boxes[b].npts(); });
    std::vector<long> load(nranks, 0);
    for (size_t q : order) {
        int best = 0;
        for (int r = 1; r < nranks; ++r) if (load[r] < load[best]) best = r;
        owner[q] = best;
        load[best] += boxes[q].npts();
    }
    return owner;
}

bool same_boxes(std::vector<BoxD> a, std::vector<BoxD> b)
{
    if (a.size() != b.size()) return false;
    auto key = [](const BoxD& x, const BoxD& y) {
        for (int d = 2; d >= 0; --d) { if (x.lo[d] != y.lo[d]) return x.lo[d] < y.lo[d]; }
        for (int d = 2; d >= 0; --d) { if (x.hi[d] != y.hi[d]) return x.hi[d] < y.hi[d]; }
        return false;
    };
    std::sort(a.begin(), a.end(), key); std::sort(b.begin(), b.end(), key);
    for (size_t q = 0; q < a.size(); ++q) for (int d = 0; d < 3; ++d) if (a[q].lo[d] != b[q].lo[d] || a[q].hi[d] != b[q].hi[d]) return false;
    return true;
}

}  // namespace

// periodic-aware erosion of a 0/1 cell map by `passes` cells (a cell survives a pass if its 26 neighbours do; outside a non-periodic
// domain face nothing constrains).  Erosion by a cube is separable and `passes` erosions by one cell are one erosion by `passes` cells:
// three 1-D minima over 2 passes + 1 cells, along x, y and z, restricted to the bounding box [lo, hi] of the cells that are set (nothing
// outside it can survive) -- round 6: the 27-neighbour loop over the whole 512^3 index space of a level cost 0.6 s per regrid above level 0
void erode_map(std::vector<unsigned char>& m, const int n[3], const int per[3], int passes, const int lo[3], const int hi[3])
{
    const int r = passes;
    const size_t sx = 1, sy = (size_t)n[0], sz = (size_t)n[0] * n[1];
    const size_t str[3] = {sx, sy, sz};
    if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]) return;
    // `o`: the map before the pass of a direction.  Only rows the pass reads are copied: x and y passes stay inside a plane, the z pass
    // reaches r planes beyond the bounding box (all planes if that wraps around).  (A copy of the whole map per direction and a range / wrap
    // test per cell in the x pass were 150 ms of every regrid above level 0 on a 512^3 index space.)
    std::unique_ptr<unsigned char[]> o(new unsigned char[m.size()]);      // (not zeroed: only what is copied below is read)
    for (int d = 0; d < 3; ++d) {
        int ka = lo[2], kb = hi[2];
        if (d == 2) { ka -= r; kb += r; if (ka < 0 || kb >= n[2]) { ka = 0; kb = n[2] - 1; } }
        if (d == 1) std::memcpy(&o[(size_t)ka * sz], &m[(size_t)ka * sz], (size_t)(kb - ka + 1) * sz);      // (the y pass reads rows j +- r: whole planes)
        else if (d == 2) std::memcpy(&o[(size_t)ka * sz], &m[(size_t)ka * sz], (size_t)(kb - ka + 1) * sz);
        else for (int k = ka; k <= kb; ++k) for (int j = lo[1]; j <= hi[1]; ++j) std::memcpy(&o[(size_t)k * sz + (size_t)j * sy], &m[(size_t)k * sz + (size_t)j * sy], (size_t)n[0]);
        for (int k = lo[2]; k <= hi[2]; ++k) for (int j = lo[1]; j <= hi[1]; ++j) {
            const int c3[3] = {0, j, k};
            for (int dd = -r; dd <= r; ++dd) {
                if (dd == 0) continue;
                if (d == 0) {
                    unsigned char* row = &m[(size_t)k * sz + (size_t)j * sy];
                    const unsigned char* orow = &o[(size_t)k * sz + (size_t)j * sy];
                    // the cells whose neighbour i + dd lies inside the row: a plain shifted AND; the few at the ends: the periodic image (or nothing)
                    const int ia = std::max(lo[0], -dd), ib = std::min(hi[0], n[0] - 1 - dd);
                    for (int i = ia; i <= ib; ++i) row[i] &= orow[i + dd];
                    if (per[0]) {
                        for (int i = lo[0]; i < ia && i <= hi[0]; ++i) row[i] &= orow[((i + dd) % n[0] + n[0]) % n[0]];
                        for (int i = std::max(ib + 1, lo[0]); i <= hi[0]; ++i) row[i] &= orow[((i + dd) % n[0] + n[0]) % n[0]];
                    }
                } else {
                    int q = c3[d] + dd;
                    if (q < 0 || q >= n[d]) { if (!per[d]) continue; q = (q % n[d] + n[d]) % n[d]; }
                    unsigned char* row = &m[(size_t)k * sz + (size_t)j * sy];
                    const unsigned char* orow = &o[(size_t)k * sz + (size_t)j * sy + ((long)q - c3[d]) * (long)str[d]];
                    for (int i = lo[0]; i <= hi[0]; ++i) row[i] &= orow[i];
                }
            }
        }
    }
}

// lbase: the levels up to lbase keep their grids (Amr::regrid(lbase)); returns the boxes of the levels lbase + 1 ...
std::vector<std::vector<BoxD>> AmrNS::make_new_grids(int lbase)
{
    ProfScope ps_prof_("regrid_make_grids");
    const int finest = (int)lev.size() - 1;
    const int max_level = rg.max_level;
    std::vector<std::vector<BoxD>> grids(max_level + 1);                 // grids[l]: boxes of level l (index space of level l), l >= 1
    const int nest_buf = 3;                                              // in cells of the level being nested (see the header comment)
    // proper nesting domain of a regrid above level 0: the new level lbase + 1 lies at least two cells of level lbase inside that level
    // (its three Godunov ghost cells and the interpolation stencil); tags of finer levels must lie two more cells inside, so that the
    // boxes they lead to, grown by the nesting buffer, still fit
    std::vector<unsigned char> allow0, allow1;
    if (lbase > 0) {
        const NavierStokes& b = *lev[lbase];
        const int nn[3] = {b.g.domain.len(0), b.g.domain.len(1), b.g.domain.len(2)};
        allow0.assign((size_t)nn[0] * nn[1] * nn[2], 0);
        int blo[3] = {nn[0], nn[1], nn[2]}, bhi[3] = {-1, -1, -1};
        for (const BoxD& bx : b.layout->boxes) {
            for (int d = 0; d < 3; ++d) { blo[d] = std::min(blo[d], bx.lo[d] - b.g.domain.lo[d]); bhi[d] = std::max(bhi[d], bx.hi[d] - b.g.domain.lo[d]); }
            for (int k = bx.lo[2]; k <= bx.hi[2]; ++k) for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                std::memset(&allow0[((size_t)(k - b.g.domain.lo[2]) * nn[1] + (j - b.g.domain.lo[1])) * nn[0] + (bx.lo[0] - b.g.domain.lo[0])], 1, (size_t)bx.len(0));
        }
        erode_map(allow0, nn, b.g.periodic, 2, blo, bhi);
        allow1 = allow0;
        erode_map(allow1, nn, b.g.periodic, 2, blo, bhi);
    }
    for (int l = std::min(finest, max_level - 1); l >= lbase; --l) {
        NavierStokes& s = *lev[l];
        const BoxD dom = s.g.domain;
        const int n0 = dom.len(0), n1 = dom.len(1), n2 = dom.len(2);
        std::unique_ptr<ProfScope> prg;
        PROF_NEXT(prg, "rg_tag_kernels");
        std::vector<unsigned char> h((size_t)n0 * n1 * n2, 0);
        // ---- NavierStokes::errorEst on the level's current data
        MultiFab tags(s.layout, cell_type(), 1, 0);
        tags.setVal(0.0);
        for (const TagRule& r : rg.rules) {
            if (l >= r.max_level || r.value.empty()) continue;
            const double v = r.value[std::min<size_t>((size_t)l, r.value.size() - 1)];
            MultiFab fld(s.layout, cell_type(), 1, 1);
            if (r.comp < 0 || r.mode == 2) {
                MultiFab vel(s.layout, cell_type(), 3, 1);
                s.fillpatch(vel, s.S[s.inew], Xvel, 3, s.bc_vel);
                derive_mag_vort(s.g, fld, 0, vel, 0);
            } else if (r.comp < 3) {
                MultiFab vel(s.layout, cell_type(), 3, 1);
                s.fillpatch(vel, s.S[s.inew], Xvel, 3, s.bc_vel);
                MultiFab::Copy(fld, vel, r.comp, 0, 1, 1);
            } else s.fillpatch(fld, s.S[s.inew], r.comp, 1, &s.bc_scal[r.comp - 3]);
            error_tag(s.g, tags, fld, 0, r.mode, v, l, r.has_box ? r.box_lo : nullptr, r.has_box ? r.box_hi : nullptr);
        }
        PROF_NEXT(prg, "rg_tags_to_host");
        for (int li = 0; li < tags.nlocal(); ++li) {
            const BoxD fb = tags.fabbox(li), vb = s.layout->lbox(li);
            std::vector<double> buf((size_t)fb.npts());
            tags.copy_to_host(li, buf.data());
            for (int k = vb.lo[2]; k <= vb.hi[2]; ++k) for (int j = vb.lo[1]; j <= vb.hi[1]; ++j) for (int i = vb.lo[0]; i <= vb.hi[0]; ++i) {
                const size_t o = ((size_t)(k - fb.lo[2]) * fb.len(1) + (j - fb.lo[1])) * fb.len(0) + (i - fb.lo[0]);
                if (buf[o] != 0.0) h[((size_t)(k - dom.lo[2]) * n1 + (j - dom.lo[1])) * n0 + (i - dom.lo[0])] = 1;
            }
        }
        // multi-rank: every rank has tagged the cells of its own boxes; the level's tag map is their union.  A cell belongs to one box,
        // hence to one rank, so the sum of the rank-local bit maps IS the union: 32 tag bits per double (exact integers), one all-reduce
        if (Context::get().comm->nranks > 1) {
            const size_t nw = (h.size() + 31) / 32;
            std::vector<double> w(nw, 0.0);
            for (size_t q = 0; q < h.size(); ++q) if (h[q]) w[q >> 5] += (double)(1u << (q & 31));
            Context::get().comm->allreduce(w.data(), (int)nw, ReduceOp::Sum);
            for (size_t q = 0; q < h.size(); ++q) h[q] = (((unsigned long long)w[q >> 5]) >> (q & 31)) & 1ull;
        }
        PROF_NEXT(prg, "rg_nesting");
        // ---- cells under the new level l+2 grids (grown by the nesting buffer at level l+1), so that level l+1 will contain them
        if (l + 2 <= max_level)
            for (const BoxD& b2 : grids[l + 2]) {
                const BoxD c = coarsen(grow(coarsen(b2, m_ratio), nest_buf), m_ratio);
                for (int k = c.lo[2]; k <= c.hi[2]; ++k) for (int j = c.lo[1]; j <= c.hi[1]; ++j) for (int i = c.lo[0]; i <= c.hi[0]; ++i) {
                    int q[3] = {i, j, k};
                    bool ok = true;
                    for (int d = 0; d < 3; ++d) {
                        if (q[d] < dom.lo[d] || q[d] > dom.hi[d]) {
                            if (!s.g.periodic[d]) { ok = false; break; }
                            q[d] = dom.lo[d] + ((q[d] - dom.lo[d]) % dom.len(d) + dom.len(d)) % dom.len(d);
                        }
                    }
                    if (ok) h[((size_t)(q[2] - dom.lo[2]) * n1 + (q[1] - dom.lo[1])) * n0 + (q[0] - dom.lo[0])] = 1;
                }
            }
        // tags outside the level's own cells cannot exist (no data there): boxes of level l+1 stay inside refine(level l) as long as
        // the clustering does not reach over the level's edge; the nesting of the OLD level l+1 in the OLD level l keeps a margin
        const int bf = std::max(1, rg.blocking_factor / m_ratio), mg = std::max(bf, rg.max_grid_size / m_ratio);
        const unsigned char* allowed = nullptr;
        std::vector<unsigned char> al;
        if (lbase > 0 && l == lbase) allowed = allow0.data();
        else if (lbase > 0) {                     // tags of a finer level: their ancestor on level lbase must lie in the inner region
            const NavierStokes& b = *lev[lbase];
            const int nb0 = b.g.domain.len(0), nb1 = b.g.domain.len(1);
            const int sh = l - lbase;
            for (int k = 0; k < n2; ++k) for (int j = 0; j < n1; ++j) for (int i = 0; i < n0; ++i) {
                const size_t q = ((size_t)k * n1 + j) * n0 + i;
                if (h[q] && !allow1[((size_t)(k >> sh) * nb1 + (j >> sh)) * nb0 + (i >> sh)]) h[q] = 0;
            }
        }
        OutflowTags oft;
        for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side)
            if (!s.g.periodic[d] && (side == 0 ? s.p.phys_lo[d] : s.p.phys_hi[d]) == phys_outflow) { oft.dir[oft.nface] = d; oft.side[oft.nface] = side; ++oft.nface; }
        oft.mode = rg.do_refine_outflow ? 1 : (rg.do_derefine_outflow ? 2 : 0);
        if (oft.mode == 2) {                      // NavierStokesBase.cpp:2147-2176 (the blocking factor and the ratio are the same on every level)
            const int np = 1;                     // Amr::nProper
            int ncc = (rg.nbuf_outflow + bf - 1) / bf, nlc = ncc * bf;
            for (int j = 1; j <= l; ++j) { nlc = nlc * m_ratio + np; ncc = (nlc + bf - 1) / bf; nlc = ncc * bf; }
            oft.ncoarse = ncc;
        }
        PROF_NEXT(prg, "rg_cluster");
        std::vector<BoxD> cb = cluster_tags(h.data(), dom, bf, mg, rg.grid_eff, rg.n_error_buf, oft.nface ? &oft : nullptr, allowed);
        for (const BoxD& b : cb) grids[l + 1].push_back(refine(b, m_ratio));
    }
    // a level can only exist if the one below it does
    for (int l = lbase + 1; l <= max_level; ++l) if (grids[l].empty()) { for (int q = l; q <= max_level; ++q) grids[q].clear(); break; }
    grids.erase(grids.begin(), grids.begin() + lbase + 1);               // -> index 0 = level lbase + 1
    while (!grids.empty() && grids.back().empty()) grids.pop_back();
    return grids;
}

// Everything that can be wrong with caller-supplied grids is found here, before the hierarchy is touched (install_grids moves the old
// levels out first: a throw half way would leave it half-built): boxes non-empty, inside the level's domain, aligned to the refinement
// ratio (a fine box covers whole coarse cells), pairwise disjoint, and every level properly nested in the one below.
void AmrNS::validate_grids(const std::vector<std::vector<BoxD>>& grids, int lbase) const
{
    BoxD dom = lev[lbase]->g.domain;
    Geometry cg = lev[lbase]->g;
    for (size_t q = 0; q < grids.size(); ++q) {
        const int l = lbase + (int)q + 1;
        if (q > 0) for (int d = 0; d < 3; ++d) { cg.domain = dom; cg.dx[d] /= (double)m_ratio; }
        dom = refine(dom, m_ratio);
        const auto& bl = grids[q];
        if (bl.empty()) throw Error("iamrx install_grids: level " + std::to_string(l) + " has no boxes");
        for (size_t a = 0; a < bl.size(); ++a) {
            const BoxD& b = bl[a];
            if (!b.ok()) throw Error("iamrx install_grids: empty box on level " + std::to_string(l));
            for (int d = 0; d < 3; ++d) {
                if (b.lo[d] < dom.lo[d] || b.hi[d] > dom.hi[d]) throw Error("iamrx install_grids: a box of level " + std::to_string(l) + " lies outside the domain");
                if (b.lo[d] % m_ratio != 0 || (b.hi[d] + 1) % m_ratio != 0)
                    throw Error("iamrx install_grids: a box of level " + std::to_string(l) + " is not aligned to the refinement ratio");
            }
            for (size_t c = 0; c < a; ++c) if (intersect(b, bl[c]).ok()) throw Error("iamrx install_grids: boxes of level " + std::to_string(l) + " overlap");
        }
        if (q > 0) check_nesting(bl, grids[q - 1], cg, l);
        else if (lbase > 0) check_nesting(bl, lev[lbase]->layout->boxes, cg, l);      // the first new level inside the level that stays
    }
}

// grids[q]: the boxes of level lbase + 1 + q; the levels up to lbase stay as they are (Amr::regrid(lbase, time)); cur_time: the time the
// levels above lbase have reached (a regrid that starts above level 0 can fall inside a coarse step)
bool AmrNS::install_grids(const std::vector<std::vector<BoxD>>& grids, int lbase, double cur_time_in)
{
    ProfScope ps_prof_("regrid_install");
    auto& ctx = Context::get();
    IAMRX_ASSERT(lbase >= 0 && lbase < (int)lev.size());
    validate_grids(grids, lbase);
    const int old_finest = (int)lev.size() - 1, new_finest = lbase + (int)grids.size();
    bool same = old_finest == new_finest;
    for (int l = lbase + 1; same && l <= new_finest; ++l) same = same_boxes(lev[l]->user_layout->boxes, grids[l - lbase - 1]);
    if (same) return false;
    ++m_grid_gen;
    const double cur_time = lbase == 0 ? lev[0]->time : cur_time_in;
    std::vector<std::unique_ptr<NavierStokes>> old;
    for (int l = lbase + 1; l <= old_finest; ++l) old.push_back(std::move(lev[l]));
    lev.resize(lbase + 1);
    lev[lbase]->fine = nullptr;
    n_cycle.resize(new_finest + 1, m_ratio); dt_level.resize(new_finest + 1, 0.0); dt_min.resize(new_finest + 1, 1.e200);
    static LayoutP empty_layout;
    if (!empty_layout) empty_layout = std::make_shared<Layout>(std::vector<BoxD>{}, std::vector<int>{}, ctx.comm->rank);
    for (int l = lbase + 1; l <= new_finest; ++l) {
        NavierStokes& c = *lev[l - 1];
        Geometry g = c.g;
        for (int d = 0; d < 3; ++d) { g.domain.lo[d] *= m_ratio; g.domain.hi[d] = (g.domain.hi[d] + 1) * m_ratio - 1; g.dx[d] /= (double)m_ratio; }
        std::unique_ptr<ProfScope> pri;
        PROF_NEXT(pri, "ri_level_object");
        LayoutP nl = std::make_shared<Layout>(grids[l - lbase - 1], distribute_boxes(grids[l - lbase - 1], ctx.comm->nranks), ctx.comm->rank);
        lev.push_back(std::make_unique<NavierStokes>(g, nl, p, o));
        NavierStokes& s = *lev.back();
        s.level = l; s.ratio = m_ratio;
        NavierStokes* ol = (l <= old_finest) ? old[l - lbase - 1].get() : nullptr;
        link_level(l);
        // ---- times (init(old): setTimeLevel(cur_time, dt_old, dt_new); init(): dt = dt_crse / ratio, dt_old = (coarse dt_old) / ratio)
        const double dt_new = ol ? dt_level[l] : dt_level[l - 1] / (double)m_ratio;
        const double dt_old = ol ? ol->st_new - ol->st_old : (c.st_new - c.st_old) / (double)m_ratio;
        dt_level[l] = dt_new; n_cycle[l] = m_ratio; dt_min[l] = ol ? dt_min[l] : 1.e200;
        s.time = cur_time; s.nstep = ol ? ol->nstep : 0; s.dt = dt_new;
        s.inew = 0; s.pnew = 0;
        s.set_time_level(cur_time, dt_old, dt_new);
        s.initial_step = false; s.initial_iter = false;
        s.profile_sections = profile_on;                           // the section profile of a run goes on across a regrid
        if (ol) for (int q = 0; q < 8; ++q) s.t_sections[q] = ol->t_sections[q];
        PROF_NEXT(pri, "ri_fill_data");
        // ---- data: FillPatch(old, S_new / P_new / Gp_new) resp. FillCoarsePatch: the old level's cells where it existed, the
        // (already rebuilt) coarser level interpolated elsewhere
        MultiFab none_c(empty_layout, cell_type(), s.nalloc, 0);
        const MultiFab* fS = ol ? &ol->S[ol->inew] : &none_c;
        TimeData fd{nullptr, fS, cur_time, cur_time};
        TimeData cd{nullptr, &c.S[c.inew], cur_time, cur_time};
        MultiFab tmp3(s.layout, cell_type(), 3, 1), tmp1(s.layout, cell_type(), 1, 1);
        fillpatch_two_levels(tmp3, 0, cur_time, fd, cd, Xvel, 3, c.g, s.g, m_ratio, s.bc_vel, s.ed_vel_lo, s.ed_vel_hi);
        MultiFab::Copy(s.S[0], tmp3, 0, Xvel, 3, 1);
        for (int q = 0; q < s.nalloc - 3; ++q) {      // the scalars, and divu / dsdt (NavierStokesBase.cpp:1742-1754, 1800-1805)
            fillpatch_two_levels(tmp1, 0, cur_time, fd, cd, Density + q, 1, c.g, s.g, m_ratio, &s.bc_scal[q], s.ed_scal_lo + 3 * q, s.ed_scal_hi + 3 * q);
            MultiFab::Copy(s.S[0], tmp1, 0, Density + q, 1, 1);
        }
        MultiFab::Copy(s.S[1], s.S[0], 0, 0, s.nalloc, 1);
        MultiFab none_g(empty_layout, cell_type(), 3, 0);
        TimeData fg{nullptr, ol ? &ol->Gp[ol->pnew] : &none_g, cur_time, cur_time};
        TimeData cg{nullptr, &c.Gp[c.pnew], cur_time, cur_time};
        fillpatch_two_levels(tmp3, 0, cur_time, fg, cg, 0, 3, c.g, s.g, m_ratio, s.bc_gp, nullptr, nullptr);
        MultiFab::Copy(s.Gp[0], tmp3, 0, 0, 3, 1);
        MultiFab::Copy(s.Gp[1], s.Gp[0], 0, 0, 3, 1);
        s.P[0].setVal(0.0);
        node_interp_from_crse(s.P[0], c.P[c.pnew], c.g, m_ratio, nullptr, false);
        if (ol) parallel_copy(s.P[0], ol->P[ol->pnew], 0, 0, 1, 0, 0, &s.g);
        MultiFab::Copy(s.P[1], s.P[0], 0, 0, 1, 1);
        s.make_rho_curr_time();
    }
    // the replaced levels: drop what the layout-keyed caches hold for them (a cached level mask keeps its layout alive otherwise)
    ProfScope ps_drop_("ri_drop_old");
    std::vector<uint64_t> dead_ids;
    for (auto& o_ : old) if (o_ && o_->layout) dead_ids.push_back(o_->layout->id);
    old.clear();                                   // the old levels' arrays go first, then what the caches hold for their layouts
    for (uint64_t id_ : dead_ids) evict_layout_caches(id_);
    return true;
}

}  // namespace iamrx
