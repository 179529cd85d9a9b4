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

namespace iamrx {

namespace {
struct PCKey {
    uint64_t dl, sl; IndexType t; int dng, sng; int per[3]; int dlo[3], dhi[3];
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

// host-only plan construction (rank `me`): every rank walks (dst box, src box, periodic image) in the same order, so the
// pack order of a sender equals the unpack order of its receiver
void build_parallel_copy_plan_host(const std::vector<BoxD>& dboxes, const std::vector<int>& downer, const std::vector<int>& dlocal_of,
                                   const std::vector<BoxD>& sboxes, const std::vector<int>& sowner, const std::vector<int>& slocal_of, int me,
                                   IndexType t, int dst_ng, int src_ng, const Geometry* pg, CopyPlan& plan, std::map<int, CopyPlan::Peer>& peers)
{
    int smin[3] = {0, 0, 0}, smax[3] = {0, 0, 0};
    if (pg) for (int d = 0; d < 3; ++d) if (pg->periodic[d]) { smin[d] = -1; smax[d] = 1; }
    for (int gd = 0; gd < (int)dboxes.size(); ++gd) {
        const bool dst_mine = downer[gd] == me;
        const BoxD dreg = grow(convert(dboxes[gd], t.t), dst_ng);
        for (int gs = 0; gs < (int)sboxes.size(); ++gs) {
            const bool src_mine = sowner[gs] == me;
            if (!dst_mine && !src_mine) continue;
            for (int sz = smin[2]; sz <= smax[2]; ++sz)
            for (int sy = smin[1]; sy <= smax[1]; ++sy)
            for (int sx = smin[0]; sx <= smax[0]; ++sx) {
                int sh[3] = {0, 0, 0};
                if (pg) { sh[0] = sx * pg->domain.len(0); sh[1] = sy * pg->domain.len(1); sh[2] = sz * pg->domain.len(2); }
                BoxD sreg = grow(convert(sboxes[gs], t.t), src_ng);
                for (int d = 0; d < 3; ++d) sreg = shift(sreg, d, sh[d]);
                const BoxD is = intersect(dreg, sreg);
                if (!is.ok()) continue;
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

void parallel_copy(MultiFab& dst, const MultiFab& src, int scomp, int dcomp, int nc, int src_ng, int dst_ng, const Geometry* periodic_geom)
{
    IAMRX_ASSERT(dst.type.t[0] == src.type.t[0] && dst.type.t[1] == src.type.t[1] && dst.type.t[2] == src.type.t[2]);
    IAMRX_ASSERT(src_ng <= src.ngrow && dst_ng <= dst.ngrow && scomp + nc <= src.ncomp && dcomp + nc <= dst.ncomp);
    static std::map<PCKey, std::unique_ptr<CopyPlan>> cache;
    PCKey key;
    std::memset(&key, 0, sizeof(key));
    key.dl = dst.layout->id; key.sl = src.layout->id; key.t = dst.type; key.dng = dst_ng; key.sng = src_ng;
    if (periodic_geom) for (int d = 0; d < 3; ++d) { key.per[d] = periodic_geom->periodic[d]; key.dlo[d] = periodic_geom->domain.lo[d]; key.dhi[d] = periodic_geom->domain.hi[d]; }
    auto it = cache.find(key);
    if (it == cache.end()) {
        auto plan = std::make_unique<CopyPlan>();
        std::map<int, CopyPlan::Peer> peers;
        const Layout &dl = *dst.layout, &sl = *src.layout;
        build_parallel_copy_plan_host(dl.boxes, dl.owner, dl.local_of, sl.boxes, sl.owner, sl.local_of, Context::get().comm->rank,
                                      dst.type, dst_ng, src_ng, periodic_geom, *plan, peers);
        plan->d_local = upload_descs(plan->local);
        for (auto& kv : peers) {
            kv.second.d_pack = upload_descs(kv.second.pack);
            kv.second.d_unpack = upload_descs(kv.second.unpack);
            plan->peers.push_back(std::move(kv.second));
            kv.second.d_pack = nullptr; kv.second.d_unpack = nullptr;
        }
        it = cache.emplace(key, std::move(plan)).first;
    }
    execute_plan(*it->second, dst, src, scomp, dcomp, nc);
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

}  // namespace iamrx
