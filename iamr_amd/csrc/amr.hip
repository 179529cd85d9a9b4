// iamr_amd/csrc/amr.hip -- inter-level data motion of the block-structured hierarchy (first building blocks):
//   parallel_copy  : amrex::MultiFab::ParallelCopy -- copy between MultiFabs that live on DIFFERENT box layouts of the same
//                    index space (intersection of every destination box with every source box, optional periodic images)
//   average_down   : amrex::average_down / average_down_faces / average_down_nodal as used by NavierStokesBase::avgDown_StatePress
//                    (reference Source/NavierStokesBase.cpp:4125-4193): fine data -> conservative average (cells), area average
//                    (faces), injection (nodes) on the coarsened fine layout, then ParallelCopy into the coarse level.
// Both run through the same cached CopyPlan machinery as the ghost exchange (one batched kernel for the local part, one
// packed message per peer).
#include "operators.h"
#include "launch.h"
#include <map>
#include <cstring>
#include <cmath>
#include <algorithm>

namespace iamrx {

namespace {
struct PCKey {
    uint64_t dl, sl; IndexType t; int dng, sng; int per[3]; int dlo[3], dhi[3]; int unique;
    bool operator<(const PCKey& o) const { return std::memcmp(this, &o, sizeof(PCKey)) < 0; }
};
CopyDesc* upload_descs(const std::vector<CopyDesc>& v)
{
    if (v.empty()) return nullptr;
    auto& ctx = Context::get();
    CopyDesc* d = (CopyDesc*)ctx.alloc(v.size() * sizeof(CopyDesc));
    IAMRX_HIP_CHECK(hipMemcpyAsync(d, v.data(), v.size() * sizeof(CopyDesc), hipMemcpyHostToDevice, ctx.stream));
    ctx.sync();
    return d;
}
}  // namespace

namespace { void box_diff(const BoxD& b, const BoxD& cut, std::vector<BoxD>& out); }   // b minus cut (below)

// host-only plan construction (rank `me`): every rank walks (dst box, src box, periodic image) in the same order, so the
// pack order of a sender equals the unpack order of its receiver
void build_parallel_copy_plan_host(const std::vector<BoxD>& dboxes, const std::vector<int>& downer, const std::vector<int>& dlocal_of,
                                   const std::vector<BoxD>& sboxes, const std::vector<int>& sowner, const std::vector<int>& slocal_of, int me,
                                   IndexType t, int dst_ng, int src_ng, const Geometry* pg, CopyPlan& plan, std::map<int, CopyPlan::Peer>& peers, bool unique)
{
    // unique (a plain copy, not an accumulation, of nodal / face data: boxes share the points on their faces): every destination point takes ONE source -- the first in the global order below -- as in
    // the FillBoundary plans (mf.hip).  Without it two descriptors of one launch wrote the same point, and where the copies differ in the
    // last bits (the two boxes' values of a shared pressure node, stale ghost cells) the result depended on which workgroup came last
    // (round 6: found when the work lists changed the order and a two-rank regrid moved by 1e-8).
    int smin[3] = {0, 0, 0}, smax[3] = {0, 0, 0};
    if (pg) for (int d = 0; d < 3; ++d) if (pg->periodic[d]) { smin[d] = -1; smax[d] = 1; }
    // (source boxes that meet a destination region through a bin index, mf.h BoxBins: the same descriptors in the same order as a scan
    // of every (dst, src, shift) triple -- source ascending, then the z, y, x shift)
    std::vector<BoxD> sregs(sboxes.size());
    for (size_t gs = 0; gs < sboxes.size(); ++gs) sregs[gs] = grow(convert(sboxes[gs], t.t), src_ng);
    const BoxBins index(sregs);
    struct Cand { int gs, sx, sy, sz; };
    std::vector<Cand> cand;
    std::vector<int> hits;
    for (int gd = 0; gd < (int)dboxes.size(); ++gd) {
        const bool dst_mine = downer[gd] == me;
        const BoxD dreg = grow(convert(dboxes[gd], t.t), dst_ng);
        cand.clear();
        for (int sz = smin[2]; sz <= smax[2]; ++sz)
        for (int sy = smin[1]; sy <= smax[1]; ++sy)
        for (int sx = smin[0]; sx <= smax[0]; ++sx) {
            BoxD q = dreg;
            if (pg) { const int s3[3] = {sx * pg->domain.len(0), sy * pg->domain.len(1), sz * pg->domain.len(2)}; for (int d = 0; d < 3; ++d) { q.lo[d] -= s3[d]; q.hi[d] -= s3[d]; } }
            index.query(q, hits);
            for (int gs : hits) cand.push_back(Cand{gs, sx, sy, sz});
        }
        std::sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) {
            if (a.gs != b.gs) return a.gs < b.gs;
            if (a.sz != b.sz) return a.sz < b.sz;
            if (a.sy != b.sy) return a.sy < b.sy;
            return a.sx < b.sx;
        });
        std::vector<BoxD> planned;
        for (const Cand& c : cand) {
            const int gs = c.gs, sx = c.sx, sy = c.sy, sz = c.sz;
            const bool src_mine = sowner[gs] == me;
            if (!unique && !dst_mine && !src_mine) continue;
            int sh[3] = {0, 0, 0};
            if (pg) { sh[0] = sx * pg->domain.len(0); sh[1] = sy * pg->domain.len(1); sh[2] = sz * pg->domain.len(2); }
            BoxD sreg = sregs[gs];
            for (int d = 0; d < 3; ++d) sreg = shift(sreg, d, sh[d]);
            const BoxD is0 = intersect(dreg, sreg);
            if (!is0.ok()) continue;
            std::vector<BoxD> parts{is0};
            if (unique) {
                for (const BoxD& q : planned) {
                    std::vector<BoxD> next;
                    for (const BoxD& pp : parts) box_diff(pp, q, next);
                    parts.swap(next);
                    if (parts.empty()) break;
                }
                for (const BoxD& pp : parts) planned.push_back(pp);
                if (!dst_mine && !src_mine) continue;          // (somebody else's pair: book-keeping only, so that every rank cuts the same pieces)
            }
            for (const BoxD& is : parts) {
                CopyDesc cd;
                cd.region = is;
                for (int d = 0; d < 3; ++d) cd.shift[d] = -sh[d];
                cd.buf_off = 0;
                if (dst_mine && src_mine) {
                    cd.src_fab = slocal_of[gs]; cd.dst_fab = dlocal_of[gd];
                    plan.local.push_back(cd);
                    plan.max_local_pts = std::max(plan.max_local_pts, is.npts());
                } else if (src_mine) {
                    auto& pr = peers[downer[gd]];
                    pr.rank = downer[gd];
                    cd.src_fab = slocal_of[gs]; cd.dst_fab = -1; cd.buf_off = pr.send_pts;
                    pr.send_pts += is.npts(); pr.max_pack_pts = std::max(pr.max_pack_pts, is.npts());
                    pr.pack.push_back(cd);
                } else {
                    auto& pr = peers[sowner[gs]];
                    pr.rank = sowner[gs];
                    cd.src_fab = -1; cd.dst_fab = dlocal_of[gd]; cd.buf_off = pr.recv_pts;
                    pr.recv_pts += is.npts(); pr.max_unpack_pts = std::max(pr.max_unpack_pts, is.npts());
                    pr.unpack.push_back(cd);
                }
            }
        }
    }
}

void parallel_copy(MultiFab& dst, const MultiFab& src, int scomp, int dcomp, int nc, int src_ng, int dst_ng, const Geometry* periodic_geom, bool add)
{
    IAMRX_ASSERT(dst.type.t[0] == src.type.t[0] && dst.type.t[1] == src.type.t[1] && dst.type.t[2] == src.type.t[2]);
    IAMRX_ASSERT(src_ng <= src.ngrow && dst_ng <= dst.ngrow && scomp + nc <= src.ncomp && dcomp + nc <= dst.ncomp);
    static std::map<PCKey, std::unique_ptr<CopyPlan>>& cache = [] () -> std::map<PCKey, std::unique_ptr<CopyPlan>>& {
        auto* c = new std::map<PCKey, std::unique_ptr<CopyPlan>>();
        register_layout_evictor([c](uint64_t lid) {
            std::vector<std::unique_ptr<CopyPlan>> dead;
            for (auto it = c->begin(); it != c->end();) { if (it->first.dl == lid || it->first.sl == lid) { dead.push_back(std::move(it->second)); it = c->erase(it); } else ++it; }
        });
        return *c;
    }();
    PCKey key;
    std::memset(&key, 0, sizeof(key));
    key.dl = dst.layout->id; key.sl = src.layout->id; key.t = dst.type; key.dng = dst_ng; key.sng = src_ng;
    // a plain copy whose sources can overlap takes one source per destination point (see build_parallel_copy_plan_host)
    // (nodal / face data only: the copies of a shared point are the same value up to rounding.  Copies that take ghost CELLS as sources --
    // the translation between a level's merged and its caller's layout -- keep every overlapping descriptor: there the copies can be
    // different things (one box's filled ghost cell, another's stale one; the valid data follow in a second pass), and "first" instead of
    // "whichever workgroup comes last" changed results by 6e-4 in tests/test_gpu_dist.py)
    const bool unique = !add && !dst.type.cell() && tune("PCOPY_UNIQUE", 1) != 0;
    key.unique = unique ? 1 : 0;
    if (periodic_geom) for (int d = 0; d < 3; ++d) { key.per[d] = periodic_geom->periodic[d]; key.dlo[d] = periodic_geom->domain.lo[d]; key.dhi[d] = periodic_geom->domain.hi[d]; }
    auto it = cache.find(key);
    if (it == cache.end()) {
        auto plan = std::make_unique<CopyPlan>();
        std::map<int, CopyPlan::Peer> peers;
        const Layout &dl = *dst.layout, &sl = *src.layout;
        build_parallel_copy_plan_host(dl.boxes, dl.owner, dl.local_of, sl.boxes, sl.owner, sl.local_of, Context::get().comm->rank,
                                      dst.type, dst_ng, src_ng, periodic_geom, *plan, peers, unique);
        plan->d_local = upload_descs(plan->local);
        for (auto& kv : peers) {
            kv.second.d_pack = upload_descs(kv.second.pack);
            kv.second.d_unpack = upload_descs(kv.second.unpack);
            plan->peers.push_back(std::move(kv.second));
            kv.second.d_pack = nullptr; kv.second.d_unpack = nullptr;
        }
        it = cache.emplace(key, std::move(plan)).first;
    }
    execute_plan(*it->second, dst, src, scomp, dcomp, nc, add);
}

// Between two layouts that hold the same cells in different boxes (a level's merged working layout and the caller's boxes, mf.h:
// coalesce_layout).  Ghost cells travel too: first everything incl. the source's ghost cells, then the valid data on top (a destination
// point covered by one box's ghost cell and another box's valid cell takes the valid one).
void relayout_copy(MultiFab& dst, const MultiFab& src, int nc, int scomp, int dcomp)
{
    const int ng = std::min(dst.ngrow, src.ngrow);
    if (dst.layout->id == src.layout->id) { MultiFab::Copy(dst, src, scomp, dcomp, nc, ng); return; }
    if (ng > 0) parallel_copy(dst, src, scomp, dcomp, nc, ng, ng, nullptr, false);
    parallel_copy(dst, src, scomp, dcomp, nc, 0, ng, nullptr, false);
}

// fine -> coarsened-fine layout: mean of the ratio^3 children (cells), of the ratio^2 coplanar children (faces), injection (nodes)
static void coarsen_onto(MultiFab& cf, const MultiFab& fine, int scomp, int ncomp, int ratio)
{
    if (cf.nlocal() == 0) return;
    const FabD *ct = cf.d_tab, *ft = fine.d_tab;
    const IndexType t = cf.type;
    const int t0 = t.t[0], t1 = t.t[1], t2 = t.t[2];
    const int r = ratio;
    // number of fine points averaged in each direction: r for a cell-like direction, 1 for a nodal one
    const int n0 = t0 ? 1 : r, n1 = t1 ? 1 : r, n2 = t2 ? 1 : r;
    const double w = 1.0 / (double)(n0 * n1 * n2);
    for_each(*cf.layout, t, 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD c = ct[f], fa = ft[f];
        for (int n = 0; n < ncomp; ++n) {
            double s = 0.0;
            for (int kr = 0; kr < n2; ++kr)
                for (int jr = 0; jr < n1; ++jr)
                    for (int ir = 0; ir < n0; ++ir) s += fa(r * i + ir, r * j + jr, r * k + kr, scomp + n);
            c(i, j, k, n) = s * w;
        }
    });
}

void average_down(const MultiFab& fine, MultiFab& crse, int scomp, int ncomp, int ratio)
{
    IAMRX_ASSERT(ratio == 2 || ratio == 4);
    LayoutP cfl = fine.layout->coarsened(ratio);
    MultiFab cf(cfl, crse.type, ncomp, 0);
    coarsen_onto(cf, fine, scomp, ncomp, ratio);
    parallel_copy(crse, cf, 0, scomp, ncomp, 0, 0, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------------
// FillPatch on a refined level (AmrLevel::FillPatch -> amrex::FillPatchTwoLevels with the StateDescriptor's interpolater,
// `cell_cons_interp` for State_Type and Gradp_Type, reference Source/NS_setup.cpp:206-394):
//   1. same-level data (time-interpolated between the old and new StateData) -> dst valid + ghost cells, periodic images incl.
//   2. the part of dst inside the (periodically extended) domain that no fine box covers is interpolated from the coarse
//      level: the coarse data (time-interpolated, physical BC applied at the coarse domain boundary) are gathered on
//      coarsen(region) grown by one cell and prolonged with amrex::CellConservativeLinear (linear limiting on):
//        central slopes (one-sided 4-point formula next to an ext_dir / hoextrap domain face), limited slope
//        s = sign(dc) min(|dc|, 2|forward|, 2|backward|) (0 at an extremum), one factor alpha_d = min_n s_n/dc_n per direction
//        shared by ALL components (the limiter stays linear in the state), fine = crse + sum_d off_d alpha_d dc_d
//   3. physical BC fill of dst at the fine domain boundary.
// The regions of step 2 and the coarse gather layout depend only on the two fine layouts and are cached.
namespace {

void box_diff(const BoxD& b, const BoxD& cut, std::vector<BoxD>& out)   // b minus cut
{
    const BoxD in = intersect(b, cut);
    if (!in.ok()) { out.push_back(b); return; }
    BoxD rem = b;
    for (int d = 2; d >= 0; --d) {
        if (rem.lo[d] < in.lo[d]) { BoxD p = rem; p.hi[d] = in.lo[d] - 1; out.push_back(p); rem.lo[d] = in.lo[d]; }
        if (rem.hi[d] > in.hi[d]) { BoxD p = rem; p.lo[d] = in.hi[d] + 1; out.push_back(p); rem.hi[d] = in.hi[d]; }
    }
}

struct FPInfo {
    LayoutP fine_patch;       // uncovered regions (fine index space), owner = owner of the dst box
    LayoutP crse_patch;       // coarsen(region) grown by 2
    std::vector<int> dst_local;   // per LOCAL patch box: local fab index of dst
    int* d_dst_local = nullptr;
    ~FPInfo() { if (d_dst_local) Context::get().free(d_dst_local); }
};

struct FPKey {
    uint64_t dl, fl; int ng, ratio; int per[3]; int dlo[3], dhi[3];
    bool operator<(const FPKey& o) const { return std::memcmp(this, &o, sizeof(FPKey)) < 0; }
};

const FPInfo& fp_info(const Layout& dl, const Layout& fl, int ng, int ratio, const Geometry& fgeom)
{
    static std::map<FPKey, std::unique_ptr<FPInfo>>& cache = [] () -> std::map<FPKey, std::unique_ptr<FPInfo>>& {
        auto* c = new std::map<FPKey, std::unique_ptr<FPInfo>>();
        register_layout_evictor([c](uint64_t lid) {
            std::vector<std::unique_ptr<FPInfo>> dead;
            for (auto it = c->begin(); it != c->end();) { if (it->first.dl == lid || it->first.fl == lid) { dead.push_back(std::move(it->second)); it = c->erase(it); } else ++it; }
        });
        return *c;
    }();
    FPKey key;
    std::memset(&key, 0, sizeof(key));
    key.dl = dl.id; key.fl = fl.id; key.ng = ng; key.ratio = ratio;
    for (int d = 0; d < 3; ++d) { key.per[d] = fgeom.periodic[d]; key.dlo[d] = fgeom.domain.lo[d]; key.dhi[d] = fgeom.domain.hi[d]; }
    auto it = cache.find(key);
    if (it != cache.end()) return *it->second;
    ProfScope ps_prof_("fp_info_build");
    auto info = std::make_unique<FPInfo>();
    const int me = Context::get().comm->rank;
    // fine valid boxes and their periodic images
    std::vector<BoxD> covered;
    int smin[3] = {0, 0, 0}, smax[3] = {0, 0, 0};
    for (int d = 0; d < 3; ++d) if (fgeom.periodic[d]) { smin[d] = -1; smax[d] = 1; }
    for (auto& b : fl.boxes)
        for (int sz = smin[2]; sz <= smax[2]; ++sz) for (int sy = smin[1]; sy <= smax[1]; ++sy) for (int sx = smin[0]; sx <= smax[0]; ++sx) {
            BoxD q = b;
            q = shift(q, 0, sx * fgeom.domain.len(0)); q = shift(q, 1, sy * fgeom.domain.len(1)); q = shift(q, 2, sz * fgeom.domain.len(2));
            covered.push_back(q);
        }
    BoxD dext = fgeom.domain;                    // cells outside it are the physical BC's business
    for (int d = 0; d < 3; ++d) if (fgeom.periodic[d]) { dext.lo[d] -= ng; dext.hi[d] += ng; }
    std::vector<BoxD> fboxes, cboxes;
    std::vector<int> owners, dst_of;
    // (the covered boxes that can cut a destination region through a bin index, mf.h BoxBins: a scan of all of them for every
    // destination box was quadratic in the number of boxes of a regridded level; same order, same pieces)
    const BoxBins cov_index(covered);
    std::vector<int> hits;
    for (int g = 0; g < (int)dl.boxes.size(); ++g) {
        std::vector<BoxD> todo{intersect(grow(dl.boxes[g], ng), dext)};
        if (!todo[0].ok()) continue;
        cov_index.query(todo[0], hits);
        for (int ci : hits) {
            const BoxD& c = covered[ci];
            std::vector<BoxD> next;
            for (auto& t : todo) box_diff(t, c, next);
            todo.swap(next);
            if (todo.empty()) break;
        }
        for (auto& u : todo) {
            fboxes.push_back(u);
            // one cell for the central slopes + one more so that the 4-point one-sided slope next to an ext_dir / hoextrap
            // domain face never has to fall back to its short form (the result then does not depend on how the uncovered region
            // happens to be chopped into boxes)
            cboxes.push_back(grow(coarsen(u, ratio), 2));
            owners.push_back(dl.owner[g]);
            dst_of.push_back(g);
        }
    }
    info->fine_patch = std::make_shared<Layout>(fboxes, owners, me);
    info->crse_patch = std::make_shared<Layout>(cboxes, owners, me);
    for (int p : info->fine_patch->local) info->dst_local.push_back(dl.local_of[dst_of[p]]);
    if (!info->dst_local.empty()) {
        auto& ctx = Context::get();
        info->d_dst_local = (int*)ctx.alloc(info->dst_local.size() * sizeof(int));
        IAMRX_HIP_CHECK(hipMemcpyAsync(info->d_dst_local, info->dst_local.data(), info->dst_local.size() * sizeof(int), hipMemcpyHostToDevice, ctx.stream));
        ctx.sync();
    }
    return *cache.emplace(key, std::move(info)).first->second;
}

// StateData time interpolation (amrex::FillPatchSingleLevel): old or new if `time` is within 1e-3 (t_new - t_old) of it
const MultiFab* time_interp(const TimeData& td, double time, int scomp, int ncomp, MultiFab& tmp, int& comp0)
{
    comp0 = scomp;
    if (!td.old_ || td.old_ == td.new_) return td.new_;
    const double eps = 1.e-3 * std::abs(td.t_new - td.t_old);
    if (std::abs(time - td.t_new) <= eps) return td.new_;
    if (std::abs(time - td.t_old) <= eps) return td.old_;
    tmp.define(td.new_->layout, td.new_->type, ncomp, 0);
    const double a = (td.t_new - time) / (td.t_new - td.t_old), b = (time - td.t_old) / (td.t_new - td.t_old);
    const FabD *tt = tmp.d_tab, *ot = td.old_->d_tab, *nt = td.new_->d_tab;
    for_each(*tmp.layout, tmp.type, 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        for (int n = 0; n < ncomp; ++n) tt[f](i, j, k, n) = a * ot[f](i, j, k, scomp + n) + b * nt[f](i, j, k, scomp + n);
    });
    comp0 = 0;
    return &tmp;
}

struct InterpBC { int dlo[3], dhi[3]; int lo[8][3], hi[8][3]; };   // coarse domain + BCType per component

// unlimited central slope of component n in direction D at coarse cell c (pointer + stride), amrex::mf_compute_slopes_{x,y,z}
__device__ __forceinline__ double cslope(const double* u, long s, int idx, int plo, int phi, int domlo, int domhi, int bclo, int bchi)
{
    double dc = 0.5 * (u[s] - u[-s]);
    if (idx == domlo && (bclo == bc_ext_dir || bclo == bc_hoextrap)) {
        if (idx + 2 <= phi) dc = -16. / 15. * u[-s] + 0.5 * u[0] + 2. / 3. * u[s] - 0.1 * u[2 * s];
        else dc = 0.25 * (u[s] + 5. * u[0] - 6. * u[-s]);
    }
    if (idx == domhi && (bchi == bc_ext_dir || bchi == bc_hoextrap)) {
        if (idx - 2 >= plo) dc = 16. / 15. * u[s] - 0.5 * u[0] - 2. / 3. * u[-s] + 0.1 * u[-2 * s];
        else dc = -0.25 * (u[-s] + 5. * u[0] - 6. * u[s]);
    }
    return dc;
}

__global__ void __launch_bounds__(256) k_cellconslin(const BoxD* __restrict__ pboxes, const int* __restrict__ dst_local, const FabD* __restrict__ dstt,
    const FabD* __restrict__ crt, int dcomp, int ncomp, int ratio, InterpBC bc)
{
    const int p = blockIdx.y;
    const BoxD b = pboxes[p];
    const FabD dst = dstt[dst_local[p]], cr = crt[p];
    const long npts = b.npts();
    const int nx = b.len(0), ny = b.len(1);
    const long s[3] = {1, (long)cr.n[0], (long)cr.n[0] * cr.n[1]};
    const double rinv = 1.0 / (double)ratio;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        const int i = b.lo[0] + (int)(q % nx), j = b.lo[1] + (int)((q / nx) % ny), k = b.lo[2] + (int)(q / ((long)nx * ny));
        const int f[3] = {i, j, k};
        int c[3];
        double off[3];
        for (int d = 0; d < 3; ++d) {
            c[d] = f[d] >= 0 ? f[d] / ratio : -((-f[d] + ratio - 1) / ratio);
            off[d] = ((double)(f[d] - c[d] * ratio) + 0.5) * rinv - 0.5;
        }
        const long co = cr.off(c[0], c[1], c[2]);
        const double exc = (double)(ratio - 1) / (double)(2 * ratio);
        for (int n = 0; n < ncomp; ++n) {
            const double* u = cr.p + co + cr.cs * n;
            const double u0 = u[0];
            double sl[3];
            for (int d = 0; d < 3; ++d) {
                const double dc = cslope(u, s[d], c[d], cr.lo[d], cr.lo[d] + cr.n[d] - 1, bc.dlo[d], bc.dhi[d], bc.lo[n][d], bc.hi[n][d]);
                const double df = 2.0 * (u[s[d]] - u0), db = 2.0 * (u0 - u[-s[d]]);
                const double sm = (df * db >= 0.0) ? fmin(fabs(df), fabs(db)) : 0.0;
                sl[d] = copysign(1.0, dc) * fmin(sm, fabs(dc));
            }
            // one factor per component: the largest excursion inside the coarse cell stays within the 27 coarse neighbours' range
            double alpha = 1.0;
            if (sl[0] != 0.0 || sl[1] != 0.0 || sl[2] != 0.0) {
                const double dumax = fabs(sl[0]) * exc + fabs(sl[1]) * exc + fabs(sl[2]) * exc;
                double umax = u0, umin = u0;
                for (int ko = -1; ko <= 1; ++ko) for (int jo = -1; jo <= 1; ++jo) for (int io = -1; io <= 1; ++io) {
                    const double v = u[io * s[0] + jo * s[1] + ko * s[2]];
                    umin = fmin(umin, v); umax = fmax(umax, v);
                }
                if (dumax * alpha > (umax - u0)) alpha = (umax - u0) / dumax;
                if (dumax * alpha > (u0 - umin)) alpha = (u0 - umin) / dumax;
            }
            dst(i, j, k, dcomp + n) = u0 + off[0] * (sl[0] * alpha) + off[1] * (sl[1] * alpha) + off[2] * (sl[2] * alpha);
        }
    }
}

}  // namespace

const MultiFab* state_time_interp(const TimeData& td, double time, int scomp, int ncomp, MultiFab& tmp, int& comp0)
{
    return time_interp(td, time, scomp, ncomp, tmp, comp0);
}

void fillpatch_two_levels(MultiFab& dst, int dcomp, double time, const TimeData& fine, const TimeData& crse, int scomp, int ncomp,
                          const Geometry& cgeom, const Geometry& fgeom, int ratio, const BCRec* bc, const double* extdir_lo, const double* extdir_hi)
{
    IAMRX_ASSERT(dst.type.cell() && ncomp <= 8 && (ratio == 2 || ratio == 4));
    auto& ctx = Context::get();
    const int ng = dst.ngrow;
    // 1. same level
    MultiFab ftmp, ctmp;
    int fc0, cc0;
    const MultiFab* fs = time_interp(fine, time, scomp, ncomp, ftmp, fc0);
    parallel_copy(dst, *fs, fc0, dcomp, ncomp, 0, ng, &fgeom);
    // 2. coarse-fine
    const FPInfo& info = fp_info(*dst.layout, *fine.new_->layout, ng, ratio, fgeom);
    if (!info.fine_patch->boxes.empty()) {
        const MultiFab* cs = time_interp(crse, time, scomp, ncomp, ctmp, cc0);
        MultiFab cmf(info.crse_patch, cell_type(), ncomp, 0);
        parallel_copy(cmf, *cs, cc0, 0, ncomp, 0, 0, &cgeom);
        bool any_wall = false;
        for (int d = 0; d < 3; ++d) any_wall = any_wall || !cgeom.periodic[d];
        if (any_wall) fill_physbc_cc(cgeom, cmf, 0, ncomp, bc, extdir_lo, extdir_hi);
        const Layout& pl = *info.fine_patch;
        if (pl.nlocal() > 0) {
            InterpBC ib;
            for (int d = 0; d < 3; ++d) {
                ib.dlo[d] = cgeom.domain.lo[d]; ib.dhi[d] = cgeom.domain.hi[d];
                for (int n = 0; n < 8; ++n) {
                    ib.lo[n][d] = (n < ncomp && !cgeom.periodic[d] && bc) ? bc[n].lo[d] : (int)bc_int_dir;
                    ib.hi[n][d] = (n < ncomp && !cgeom.periodic[d] && bc) ? bc[n].hi[d] : (int)bc_int_dir;
                }
            }
            long maxpts = 0;
            for (int li = 0; li < pl.nlocal(); ++li) maxpts = std::max(maxpts, pl.lbox(li).npts());
            const unsigned gx = (unsigned)std::min<long>((maxpts + 255) / 256, 1024);
            hipLaunchKernelGGL(k_cellconslin, dim3(gx, (unsigned)pl.nlocal()), dim3(256), 0, ctx.stream, pl.d_boxes, info.d_dst_local, dst.d_tab,
                               cmf.d_tab, dcomp, ncomp, ratio, ib);
        }
    }
    // 3. physical BC at the fine domain boundary
    bool any_wall = false;
    for (int d = 0; d < 3; ++d) any_wall = any_wall || !fgeom.periodic[d];
    if (any_wall) fill_physbc_cc(fgeom, dst, dcomp, ncomp, bc, extdir_lo, extdir_hi);
}

// ------------------------------------------------------------------------------------------------------------------------
// Flux register of a coarse-fine interface (amrex::FluxRegister / YAFluxRegister as IAMR uses them: advective registers in
// NavierStokesBase::ComputeAofs, reference Source/NavierStokesBase.cpp:5036-5096; viscous registers in
// NavierStokes::scalar_diffusion_update / Diffusion::diffuse_tensor_velocity, Source/NavierStokes.cpp:975-992,
// Source/Diffusion.cpp:940-953; consumed by NavierStokes::reflux, Source/NavierStokes.cpp:1736-1838).
// One register value per coarse face on the boundary of every (coarsened) fine box, per component:
//   CrseInit / CrseAdd : reg  = / += mult * coarse flux          (IAMR: mult = -dt_crse, fluxes already area-weighted)
//   FineAdd            : reg += mult * sum of the ratio^2 fine fluxes on that coarse face   (mult = +dt_fine)
//   Reflux             : S(coarse cell outside the fine box) -= / += scale * reg / volume   (low / high side of the fine box)
// Storage: per direction and side one face-type MultiFab on the layout of one-cell-thick slabs of OUTSIDE coarse cells
// (owner = owner of the fine box); the register lives on the slab's face that touches the fine box.  Reflux adds the six
// slab sets one after the other (slabs of one set are disjoint), with the periodic images of the coarse domain.
FluxRegister::FluxRegister(LayoutP fine, LayoutP crse, const Geometry& cgeom, int ratio, int ncomp)
    : m_fine(std::move(fine)), m_crse(std::move(crse)), m_cgeom(cgeom), m_ratio(ratio), m_ncomp(ncomp)
{
    for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
            std::vector<BoxD> slabs;
            for (auto& b : m_fine->boxes) {
                BoxD cb = coarsen(b, ratio);
                BoxD s = cb;
                if (side == 0) { s.lo[d] = s.hi[d] = cb.lo[d] - 1; } else { s.lo[d] = s.hi[d] = cb.hi[d] + 1; }
                slabs.push_back(s);
            }
            m_slab[d][side] = std::make_shared<Layout>(slabs, m_fine->owner, Context::get().comm->rank);
            m_reg[d][side].define(m_slab[d][side], face_type(d), ncomp, 0);
            m_reg[d][side].setVal(0.0);
        }
}

void FluxRegister::setVal(double v)
{
    for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) m_reg[d][s].setVal(v);
}

void FluxRegister::CrseInit(const MultiFab& flux, int dir, int scomp, int dcomp, int nc, double mult, bool add)
{
    IAMRX_ASSERT(flux.type.t[dir] == 1 && dcomp + nc <= m_ncomp);
    for (int side = 0; side < 2; ++side) {
        MultiFab& reg = m_reg[dir][side];
        MultiFab tmp(m_slab[dir][side], face_type(dir), nc, 0);
        tmp.setVal(0.0);
        parallel_copy(tmp, flux, scomp, 0, nc, 0, 0, nullptr);
        const FabD *rt = reg.d_tab, *tt = tmp.d_tab;
        const BoxD* sb = m_slab[dir][side]->d_boxes;
        const int plane_off = side == 0 ? 1 : 0;       // the slab's face that touches the fine box
        for_each(*m_slab[dir][side], cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
            int q[3] = {i, j, k};
            q[dir] = (dir == 0 ? sb[f].lo[0] : (dir == 1 ? sb[f].lo[1] : sb[f].lo[2])) + plane_off;
            for (int n = 0; n < nc; ++n) {
                const double v = mult * tt[f](q[0], q[1], q[2], n);
                if (add) rt[f](q[0], q[1], q[2], dcomp + n) += v; else rt[f](q[0], q[1], q[2], dcomp + n) = v;
            }
        });
    }
}

void FluxRegister::FineAdd(const MultiFab& flux, int dir, int scomp, int dcomp, int nc, double mult)
{
    IAMRX_ASSERT(flux.type.t[dir] == 1 && flux.layout->id == m_fine->id && dcomp + nc <= m_ncomp);
    const int r = m_ratio;
    const int d1 = dir == 0 ? 1 : 0, d2 = dir == 2 ? 1 : 2;
    for (int side = 0; side < 2; ++side) {
        MultiFab& reg = m_reg[dir][side];
        const FabD *rt = reg.d_tab, *ft = flux.d_tab;
        const BoxD* sb = m_slab[dir][side]->d_boxes;
        const int plane_off = side == 0 ? 1 : 0;
        for_each(*m_slab[dir][side], cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
            int q[3] = {i, j, k};
            q[dir] = (dir == 0 ? sb[f].lo[0] : (dir == 1 ? sb[f].lo[1] : sb[f].lo[2])) + plane_off;     // coarse face index
            for (int n = 0; n < nc; ++n) {
                double s = 0.0;
                for (int b = 0; b < r; ++b)
                    for (int a = 0; a < r; ++a) {
                        int p[3];
                        p[dir] = r * q[dir]; p[d1] = r * q[d1] + a; p[d2] = r * q[d2] + b;
                        s += ft[f](p[0], p[1], p[2], scomp + n);
                    }
                rt[f](q[0], q[1], q[2], dcomp + n) += mult * s;
            }
        });
    }
}

void FluxRegister::Reflux(MultiFab& S, double volume, double scale, int scomp, int dcomp, int nc)
{
    IAMRX_ASSERT(S.type.cell() && S.layout->id == m_crse->id);
    for (int dir = 0; dir < 3; ++dir)
        for (int side = 0; side < 2; ++side) {
            MultiFab tmp(m_slab[dir][side], cell_type(), nc, 0);
            const FabD *rt = m_reg[dir][side].d_tab, *tt = tmp.d_tab;
            const int plane_off = side == 0 ? 1 : 0;
            const double m = (side == 0 ? -scale : scale) / volume;     // FluxRegister::Reflux: low side of the fine box: -scale
            for_each(*m_slab[dir][side], cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
                int q[3] = {i, j, k};
                q[dir] += plane_off;
                for (int n = 0; n < nc; ++n) tt[f](i, j, k, n) = m * rt[f](q[0], q[1], q[2], scomp + n);
            });
            parallel_copy(S, tmp, 0, dcomp, nc, 0, 0, &m_cgeom, true);
        }
}

// ------------------------------------------------------------------------------------------------------------------------
// NavierStokesBase::create_umac_grown on a refined level (reference Source/NavierStokesBase.cpp:1108-1311):
//   1. ghost faces of the fine mac velocities: amrex::FillPatchTwoLevels with the FaceLinear interpolater (linear between the
//      two coarse faces in the face-normal direction, piecewise constant in the transverse ones), fine data where a fine box
//      (or its periodic image) covers them;
//   2. IAMR's own divergence fix: a ghost CELL that is not covered by fine data and has exactly one face neighbour inside the
//      fine level gets the outer face (w.r.t. that neighbour) reset so that div(u_mac) = divu in it; grid edges / corners
//      are left alone.  Level mask values as in the reference: interior 0, covered 1, notcovered 2, physbnd 3.
namespace {

struct MaskKey { uint64_t fl; int per[3]; int dlo[3], dhi[3]; bool operator<(const MaskKey& o) const { return std::memcmp(this, &o, sizeof(MaskKey)) < 0; } };

// the level mask (2 ghost cells), kept as doubles in a cell-centred MultiFab
const MultiFab& level_mask(const LayoutP& fl, const Geometry& fgeom)
{
    static std::map<MaskKey, std::unique_ptr<MultiFab>>& cache = [] () -> std::map<MaskKey, std::unique_ptr<MultiFab>>& {
        auto* c = new std::map<MaskKey, std::unique_ptr<MultiFab>>();
        // the mask holds its layout: only an explicit evict_layout_caches (regrid) reaches this entry
        register_layout_evictor([c](uint64_t lid) {
            std::vector<std::unique_ptr<MultiFab>> dead;
            for (auto it = c->begin(); it != c->end();) { if (it->first.fl == lid) { dead.push_back(std::move(it->second)); it = c->erase(it); } else ++it; }
        });
        return *c;
    }();
    MaskKey key;
    std::memset(&key, 0, sizeof(key));
    key.fl = fl->id;
    for (int d = 0; d < 3; ++d) { key.per[d] = fgeom.periodic[d]; key.dlo[d] = fgeom.domain.lo[d]; key.dhi[d] = fgeom.domain.hi[d]; }
    auto it = cache.find(key);
    if (it != cache.end()) return *it->second;
    auto m = std::make_unique<MultiFab>(fl, cell_type(), 1, 2);
    const BoxD dom = fgeom.domain;
    const int p0 = fgeom.periodic[0], p1 = fgeom.periodic[1], p2 = fgeom.periodic[2];
    const FabD* mt = m->d_tab;
    for_each(*fl, cell_type(), 2, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const bool in = (p0 || (i >= dom.lo[0] && i <= dom.hi[0])) && (p1 || (j >= dom.lo[1] && j <= dom.hi[1])) && (p2 || (k >= dom.lo[2] && k <= dom.hi[2]));
        mt[f](i, j, k) = in ? 2.0 : 3.0;                                   // notcovered : physbnd
    });
    MultiFab ones(fl, cell_type(), 1, 0);
    ones.setVal(1.0);
    parallel_copy(*m, ones, 0, 0, 1, 0, 2, &fgeom);                        // covered (own valid region included for now)
    for_each(*fl, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) { mt[f](i, j, k) = 0.0; });   // interior
    return *cache.emplace(key, std::move(m)).first->second;
}

}  // namespace

void create_umac_grown(MultiFab* const umac_fine[3], const MultiFab* const umac_crse[3], const MultiFab* divu,
                       const Geometry& cgeom, const Geometry& fgeom, int ratio)
{
    const LayoutP fl = umac_fine[0]->layout;
    IAMRX_ASSERT(umac_fine[0]->ngrow >= 1 && (ratio == 2 || ratio == 4));
    auto& ctx = Context::get();
    // coarse faces under every fine box grown by one cell
    std::vector<BoxD> cb;
    for (auto& b : fl->boxes) cb.push_back(coarsen(grow(b, 1), ratio));
    static std::map<std::pair<uint64_t, int>, LayoutP>& cl_cache = [] () -> std::map<std::pair<uint64_t, int>, LayoutP>& {
        auto* c = new std::map<std::pair<uint64_t, int>, LayoutP>();
        register_layout_evictor([c](uint64_t lid) {
            std::vector<LayoutP> dead;
            for (auto it = c->begin(); it != c->end();) { if (it->first.first == lid) { dead.push_back(std::move(it->second)); it = c->erase(it); } else ++it; }
        });
        return *c;
    }();
    LayoutP& cl = cl_cache[{fl->id, ratio}];
    if (!cl) cl = std::make_shared<Layout>(cb, fl->owner, ctx.comm->rank);
    for (int d = 0; d < 3; ++d) {
        MultiFab& uf = *umac_fine[d];
        MultiFab cpatch(cl, face_type(d), 1, 0);
        cpatch.setVal(0.0);
        parallel_copy(cpatch, *umac_crse[d], 0, 0, 1, 0, 0, &cgeom);
        const FabD *ft = uf.d_tab, *ct = cpatch.d_tab;
        const BoxD* vb = fl->d_boxes;
        const int r = ratio;
        const double rinv = 1.0 / (double)ratio;
        // every ghost face (faces of the grown box that are not faces of the valid box) from the coarse level: FaceLinear
        for_each(*fl, face_type(d), 1, ctx.stream, [=] __device__(int i, int j, int k, int f) {
            const BoxD b = vb[f];
            const int fi[3] = {i, j, k};
            bool valid = true;
            for (int e = 0; e < 3; ++e) valid = valid && fi[e] >= b.lo[e] && fi[e] <= b.hi[e] + (e == d ? 1 : 0);
            if (valid) return;
            int c[3];
            for (int e = 0; e < 3; ++e) c[e] = fi[e] >= 0 ? fi[e] / r : -((-fi[e] + r - 1) / r);
            const FabD cf = ct[f];
            const int rem = fi[d] - c[d] * r;
            double v;
            if (rem == 0) v = cf(c[0], c[1], c[2]);
            else {
                const double w = (double)rem * rinv;
                int cp[3] = {c[0], c[1], c[2]};
                cp[d] += 1;
                v = (1.0 - w) * cf(c[0], c[1], c[2]) + w * cf(cp[0], cp[1], cp[2]);
            }
            ft[f](i, j, k) = v;
        });
        uf.FillBoundary(fgeom);          // ghost faces covered by other fine boxes (or periodic images) take the fine data
    }
    // divergence fix in the not-covered ghost cells with exactly one face neighbour inside the level
    const MultiFab& mask = level_mask(fl, fgeom);
    const FabD *mt = mask.d_tab, *ut = umac_fine[0]->d_tab, *vt = umac_fine[1]->d_tab, *wt = umac_fine[2]->d_tab;
    const FabD* dt_ = divu ? divu->d_tab : nullptr;
    const BoxD* vb = fl->d_boxes;
    const double dx0 = fgeom.dx[0], dx1 = fgeom.dx[1], dx2 = fgeom.dx[2];
    const double dxi0 = 1.0 / dx0, dxi1 = 1.0 / dx1, dxi2 = 1.0 / dx2;
    for_each(*fl, cell_type(), 1, ctx.stream, [=] __device__(int i, int j, int k, int f) {
        const FabD m = mt[f];
        if (m(i, j, k) != 2.0) return;
        int count = 0;
        count += (m(i - 1, j, k) == 0.0 || m(i - 1, j, k) == 1.0); count += (m(i + 1, j, k) == 0.0 || m(i + 1, j, k) == 1.0);
        count += (m(i, j - 1, k) == 0.0 || m(i, j - 1, k) == 1.0); count += (m(i, j + 1, k) == 0.0 || m(i, j + 1, k) == 1.0);
        count += (m(i, j, k - 1) == 0.0 || m(i, j, k - 1) == 1.0); count += (m(i, j, k + 1) == 0.0 || m(i, j, k + 1) == 1.0);
        if (count != 1) return;
        const BoxD b = vb[f];
        const FabD u = ut[f], v = vt[f], w = wt[f];
        const double div = dt_ ? dt_[f](i, j, k) : 0.0;
        const double dux = dxi0 * (u(i + 1, j, k) - u(i, j, k));
        const double duy = dxi1 * (v(i, j + 1, k) - v(i, j, k));
        const double duz = dxi2 * (w(i, j, k + 1) - w(i, j, k));
        if (i < b.lo[0] && m(i + 1, j, k) != 2.0) u(i, j, k) = u(i + 1, j, k) + dx0 * (duy + duz - div);
        else if (i > b.hi[0] && m(i - 1, j, k) != 2.0) u(i + 1, j, k) = u(i, j, k) - dx0 * (duy + duz - div);
        if (j < b.lo[1] && m(i, j + 1, k) != 2.0) v(i, j, k) = v(i, j + 1, k) + dx1 * (dux + duz - div);
        else if (j > b.hi[1] && m(i, j - 1, k) != 2.0) v(i, j + 1, k) = v(i, j, k) - dx1 * (dux + duz - div);
        if (k < b.lo[2] && m(i, j, k + 1) != 2.0) w(i, j, k) = w(i, j, k + 1) + dx2 * (dux + duy - div);
        else if (k > b.hi[2] && m(i, j, k - 1) != 2.0) w(i, j, k + 1) = w(i, j, k) - dx2 * (dux + duy - div);
    });
}

}  // namespace iamrx
