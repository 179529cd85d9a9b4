// iamr_amd/csrc/launch.h -- level-wide launch geometry for gfx950.
//
// One launch covers every local FAB of a level: blockIdx.y = local fab, blockIdx.x = tile.
// A workgroup is 256 threads = 4 wavefronts of 64; a tile is BX x BY x TZ index points with
// BX*BY = 256, BX a power of two <= 64 so that each wavefront reads one contiguous x-row segment
// (coalesced 512 B per row for BX = 64).  Each thread marches TZ points in z.
#pragma once
#include "core.h"
#include "mf.h"
#include <map>
#include <array>
#include <tuple>
#include <vector>

namespace iamrx {

struct Tiling {
    int bxs;            // log2(BX)
    int ntx, nty, ntz;  // tiles per direction (for the largest local box)
    int tz;             // z points per thread
    int nfab;
    int xcd_cnt;        // > 0: XCD-aware order, tiles per XCD (see tile_ijk)
    // flat tile list (level_tiling with allow_list on a level whose boxes differ in size): workgroup b takes entry b = (fab, i0, j0,
    // k0 | log2(BX) << 26) -- offsets inside the box -- instead of tile b of box blockIdx.y in a grid sized for the LARGEST box
    const int4* list = nullptr;
    int nlist = 0;
    dim3 grid() const
    {
        if (list) return dim3((unsigned)nlist, 1, 1);
        return dim3((unsigned)(xcd_cnt > 0 ? 8 * xcd_cnt : ntx * nty * ntz), (unsigned)(nfab > 0 ? nfab : 1), 1);
    }
    static dim3 block() { return dim3(256, 1, 1); }
};

inline Tiling make_tiling(const int maxlen[3], int nfab, int tz = 4)
{
    Tiling t;
    int bx = 64, bxs = 6;
    while (bx > 4 && bx / 2 >= maxlen[0]) { bx /= 2; --bxs; }
    int by = 256 / bx;
    t.bxs = bxs;
    t.ntx = (maxlen[0] + bx - 1) / bx;
    t.nty = (maxlen[1] + by - 1) / by;
    // small levels: a workgroup that marches tz planes pays their memory latencies one after the other while most of the chip idles
    // (an 8^3 multigrid level was ONE workgroup with 32 active threads, 10 us per colour pass); give the planes to more workgroups
    // until the launch holds ~512 of them
    const int tz_adapt = (int)tune("TZ_ADAPT", 1);
    while (tz_adapt && tz > 1 && (long)t.ntx * t.nty * (nfab > 0 ? nfab : 1) * ((maxlen[2] + tz - 1) / tz) < 512) tz /= 2;
    t.tz = tz;
    t.ntz = (maxlen[2] + tz - 1) / tz;
    t.nfab = nfab;
    // Workgroup b of a launch runs on XCD b % 8 and every XCD has its own L2.  With the plain order the y/z-neighbours of a tile
    // run on other XCDs, so the stencil halo rows shared by neighbouring tiles are fetched from HBM once per XCD.  XCD-aware
    // order: XCD q works through the contiguous tile range [q*cnt, (q+1)*cnt) (a z-slab of the box), neighbouring tiles then
    // meet in the same L2.  Speed only -- every tile is still processed exactly once.
    const int xcd_on = (int)tune("XCD_TILES", 1);
    const int total = t.ntx * t.nty * t.ntz;
    t.xcd_cnt = (xcd_on && total >= 64) ? (total + 7) / 8 : 0;
    return t;
}

// (mf.hip) cached flat tile list of the local boxes of l (converted to type, grown by ng), tz planes per thread; *total = its length
const int4* level_tile_list(const Layout& l, const IndexType& type, int ng, int tz, int* total, int xdiv = 1);
// (mf.hip) any other device-resident int4 work list that belongs to a layout: built once per (layout, subkey), freed with the layout
const int4* layout_int4_list(const Layout& l, const std::array<long, 5>& subkey, const std::function<void(std::vector<int4>&)>& build, int* total);

// tiling for the local boxes of `l` converted to `type` and grown by ng.  allow_list: the kernel takes its box from tile_fab(t), so a level
// whose boxes differ in size (a regridded refined level and its multigrid levels: boxes of 2 ... 32 cells side by side) may be given a flat
// list of the tiles that exist -- the grid of the plain form is sized for the largest box, and on such levels most of its workgroups
// find nothing to do (round 6: the Krylov iterations on the coarsest level of BASELINE config C5, 3 x per launch)
inline Tiling level_tiling(const Layout& l, const IndexType& type, int ng, int tz = 4, bool allow_list = false)
{
    int ml[3];
    for (int d = 0; d < 3; ++d) ml[d] = l.max_len[d] + type.t[d] + 2 * ng;
    Tiling t = make_tiling(ml, l.nlocal(), tz);
    if (allow_list && l.nlocal() >= 4 && tune("TILE_LISTS", 1) != 0) {
        int n = 0;
        const int4* lst = level_tile_list(l, type, ng, t.tz, &n);
        const long plain = (long)t.ntx * t.nty * t.ntz * l.nlocal();
        if (lst && 4L * n <= 3L * plain) { t.list = lst; t.nlist = n; t.xcd_cnt = 0; }
    }
    return t;
}

// Boundary-region descriptor lists (which ghost slabs of which box lie outside the domain / on a wall) depend only on the layout, the
// ghost width and the domain: built on the host and uploaded ONCE per key, then reused by every later call (a multigrid level calls
// its BC fills for every colour pass).  The device copies live for the life of the process (a few KB per layout).
// descriptor cache of one call site: key[0] is the layout id; never destroyed, entries leave with their layout
template <class D>
inline std::map<std::array<long, 10>, std::tuple<D*, int, long>>& make_desc_cache()
{
    auto* c = new std::map<std::array<long, 10>, std::tuple<D*, int, long>>();
    register_layout_evictor([c](uint64_t lid) {
        for (auto it = c->begin(); it != c->end();) {
            if ((uint64_t)it->first[0] == lid) { if (std::get<0>(it->second)) (void)hipFree(std::get<0>(it->second)); it = c->erase(it); } else ++it;
        }
    });
    return *c;
}

template <class D, class F>
inline const D* cached_descs(std::map<std::array<long, 10>, std::tuple<D*, int, long>>& cache, const std::array<long, 10>& key, F build,
                             int& n, long& maxpts)
{
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<D> v;
        long mp = 0;
        build(v, mp);
        D* d = nullptr;
        if (!v.empty()) {
            IAMRX_HIP_CHECK(hipMalloc(&d, v.size() * sizeof(D)));
            IAMRX_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(D), hipMemcpyHostToDevice));
        }
        it = cache.emplace(key, std::make_tuple(d, (int)v.size(), mp)).first;
    }
    n = std::get<1>(it->second);
    maxpts = std::get<2>(it->second);
    return std::get<0>(it->second);
}

// the same for kernels that index a box through cell PAIRS in x (k_abec_gsrb2: tile space = the box with (len + 1) / 2 columns)
inline Tiling pair_tiling(const Layout& l, int tz)
{
    int ml[3] = {(l.max_len[0] + 1) / 2, l.max_len[1], l.max_len[2]};
    Tiling t = make_tiling(ml, l.nlocal(), tz);
    if (l.nlocal() >= 4 && tune("TILE_LISTS", 1) != 0) {
        int n = 0;
        const int4* lst = level_tile_list(l, IndexType{{0, 0, 0}}, 0, t.tz, &n, 2);
        const long plain = (long)t.ntx * t.nty * t.ntz * l.nlocal();
        if (lst && 4L * n <= 3L * plain) { t.list = lst; t.nlist = n; t.xcd_cnt = 0; }
    }
    return t;
}

#ifdef __HIPCC__
// decode (i, j, k-range) of this thread inside box b; returns false if (i,j) is outside
// the box of this workgroup (kernels launched with a tiling that may carry a flat list) and log2 of its row length in threads
__device__ __forceinline__ int tile_fab(const Tiling& t) { return t.list ? t.list[blockIdx.x].x : (int)blockIdx.y; }
__device__ __forceinline__ int tile_bxs(const Tiling& t) { return t.list ? (int)((unsigned)t.list[blockIdx.x].w >> 26) : t.bxs; }

__device__ __forceinline__ bool tile_ijk(const Tiling& t, const BoxD& b, int& i, int& j, int& k0, int& k1)
{
    if (t.list) {
        const int4 e = t.list[blockIdx.x];
        const int bxs = (int)((unsigned)e.w >> 26), tid = threadIdx.x;
        i = b.lo[0] + e.y + (tid & ((1 << bxs) - 1));
        j = b.lo[1] + e.z + (tid >> bxs);
        k0 = b.lo[2] + (e.w & 0x3ffffff);
        k1 = k0 + t.tz - 1;
        if (k1 > b.hi[2]) k1 = b.hi[2];
        return i <= b.hi[0] && j <= b.hi[1] && k0 <= b.hi[2];
    }
    const int bx = 1 << t.bxs;
    const int tid = threadIdx.x;
    const int tx = tid & (bx - 1), ty = tid >> t.bxs;
    int bid = blockIdx.x;
    if (t.xcd_cnt > 0) {
        bid = (bid & 7) * t.xcd_cnt + (bid >> 3);
        if (bid >= t.ntx * t.nty * t.ntz) return false;
    }
    const int bxi = bid % t.ntx;
    const int r = bid / t.ntx;
    const int byi = r % t.nty, bzi = r / t.nty;
    i = b.lo[0] + bxi * bx + tx;
    j = b.lo[1] + byi * (256 >> t.bxs) + ty;
    k0 = b.lo[2] + bzi * t.tz;
    k1 = k0 + t.tz - 1;
    if (k1 > b.hi[2]) k1 = b.hi[2];
    return i <= b.hi[0] && j <= b.hi[1] && k0 <= b.hi[2];
}

__device__ __forceinline__ BoxD dev_grow_convert(BoxD b, int t0, int t1, int t2, int ng)
{
    b.lo[0] -= ng; b.lo[1] -= ng; b.lo[2] -= ng;
    b.hi[0] += t0 + ng; b.hi[1] += t1 + ng; b.hi[2] += t2 + ng;
    return b;
}

// generic level-wide loop: f(i,j,k,fab) for every point of (valid box converted to type, grown ng)
template <class F>
__global__ void __launch_bounds__(256) k_for_each(Tiling t, const BoxD* __restrict__ boxes, int t0, int t1, int t2, int ng, F f)
{
    const int fab = tile_fab(t);
    const BoxD b = dev_grow_convert(boxes[fab], t0, t1, t2, ng);
    int i, j, k0, k1;
    if (!tile_ijk(t, b, i, j, k0, k1)) return;
    for (int k = k0; k <= k1; ++k) f(i, j, k, fab);
}

template <class F>
inline void for_each(const Layout& l, const IndexType& type, int ng, hipStream_t s, F f)
{
    if (l.nlocal() == 0) return;
    Tiling t = level_tiling(l, type, ng, 4, true);
    hipLaunchKernelGGL((k_for_each<F>), t.grid(), Tiling::block(), 0, s, t, l.d_boxes, type.t[0], type.t[1], type.t[2], ng, f);
}

// maxima of NOUT non-negative quantities that a functor forms at every point -- f(i, j, k, fab, m) raises m[0 .. NOUT) -- in one pass
// over the level and one read-back (no intermediate arrays; out[n] is the maximum over all ranks unless the layout is replicated)
void reduce_finish_max(int nout, int np, bool global, double* out);      // k_basic.hip: block partials -> out (host)
template <int NOUT, class F>
__global__ void __launch_bounds__(256) k_reduce_max_f(Tiling t, const BoxD* __restrict__ boxes, int t0, int t1, int t2, int ng, F f,
                                                      double* __restrict__ partials, int np)
{
    const int fab = tile_fab(t);
    const BoxD b = dev_grow_convert(boxes[fab], t0, t1, t2, ng);
    int i, j, k0, k1;
    double m[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) m[n] = 0.0;
    if (tile_ijk(t, b, i, j, k0, k1)) for (int k = k0; k <= k1; ++k) f(i, j, k, fab, m);
    __shared__ double sm[NOUT][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        double v = m[n];
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(v, off, 64); v = o > v ? o : v; }
        if (lane == 0) sm[n][w] = v;
    }
    __syncthreads();
    if (threadIdx.x < NOUT) {
        const int n = threadIdx.x;
        double v = sm[n][0];
        for (int q = 1; q < 4; ++q) v = sm[n][q] > v ? sm[n][q] : v;
        partials[(size_t)n * np + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = v;
    }
}

template <int NOUT, class F>
inline void reduce_max_f(const Layout& l, const IndexType& type, int ng, F f, double* out, bool global = true)
{
    auto& ctx = Context::get();
    int np = 0;
    if (l.nlocal() > 0) {
        Tiling t = level_tiling(l, type, ng, 8, true);
        dim3 g = t.grid();
        np = (int)(g.x * g.y);
        ctx.ensure_scratch((size_t)NOUT * np + 16);
        hipLaunchKernelGGL((k_reduce_max_f<NOUT, F>), g, Tiling::block(), 0, ctx.stream, t, l.d_boxes, type.t[0], type.t[1], type.t[2], ng, f, ctx.d_scratch, np);
    }
    reduce_finish_max(NOUT, np, global && !l.replicated, out);
}

// the same with the maxima left on the device at d_out (k_basic.hip reduce_finish_dev; no read-back); f may write arrays as a side effect
void reduce_finish_dev(int op, int nout, int np, bool global, double* d_out);
template <int NOUT, class F>
inline void reduce_max_f_dev(const Layout& l, const IndexType& type, int ng, F f, double* d_out, bool global = true)
{
    auto& ctx = Context::get();
    int np = 0;
    if (l.nlocal() > 0) {
        Tiling t = level_tiling(l, type, ng, 8, true);
        dim3 g = t.grid();
        np = (int)(g.x * g.y);
        ctx.ensure_scratch((size_t)NOUT * np + 16);
        hipLaunchKernelGGL((k_reduce_max_f<NOUT, F>), g, Tiling::block(), 0, ctx.stream, t, l.d_boxes, type.t[0], type.t[1], type.t[2], ng, f, ctx.d_scratch, np);
    }
    reduce_finish_dev(1, NOUT, np, global && !l.replicated, d_out);
}

// level-wide loop in two phases, for the BLAS-1 style operations: v = ld(i, j, k, fab, n) for NP planes -- every load issued before the
// first store (destination and sources may be the same array, so a plain loop orders each load behind the previous store) -- then
// st(i, j, k, fab, n, v)
template <int NP, class L, class S>
__global__ void __launch_bounds__(256) k_for_each_2ph(Tiling t, const BoxD* __restrict__ boxes, int t0, int t1, int t2, int ng, int nc, L ld, S st)
{
    const int fab = tile_fab(t);
    const BoxD b = dev_grow_convert(boxes[fab], t0, t1, t2, ng);
    int i, j, k0, k1;
    if (!tile_ijk(t, b, i, j, k0, k1)) return;
    for (int n = 0; n < nc; ++n)
        for (int k = k0; k <= k1; k += NP) {
            double v[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) v[p] = k + p <= k1 ? ld(i, j, k + p, fab, n) : 0.0;
#pragma unroll
            for (int p = 0; p < NP; ++p) if (k + p <= k1) st(i, j, k + p, fab, n, v[p]);
        }
}

template <class L, class S>
inline void for_each_2ph(const Layout& l, const IndexType& type, int ng, int nc, hipStream_t s, L ld, S st)
{
    if (l.nlocal() == 0 || nc <= 0) return;
    Tiling t = level_tiling(l, type, ng, 4, true);
    hipLaunchKernelGGL((k_for_each_2ph<4, L, S>), t.grid(), Tiling::block(), 0, s, t, l.d_boxes, type.t[0], type.t[1], type.t[2], ng, nc, ld, st);
}
#endif

}  // namespace iamrx
