// iamr_amd/csrc/k_basic.hip -- streaming kernels: fill, plan copies (ghost exchange), BLAS-1 style
// fused ops and deterministic two-stage reductions (block partials -> single-block finish).
// Role: amrex MultiFab::{setVal,Copy,Saxpy,Xpay,mult,norm0}, FillBoundary pack/unpack.
#include "kernels.h"
#include "launch.h"

namespace iamrx {

__global__ void __launch_bounds__(256) k_fill(double* __restrict__ p, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = v;
}

void launch_fill(double* p, size_t n, double v, hipStream_t s)
{
    if (n == 0) return;
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)nb), dim3(256), 0, s, p, n, v);
}

// one blockIdx.y per copy descriptor; threads stride over region points (x fastest => coalesced rows)
__global__ void __launch_bounds__(256) k_copy_plan(const CopyDesc* __restrict__ descs, const FabD* __restrict__ src,
                                                   const FabD* __restrict__ dst, int scomp, int dcomp, int nc, int add)
{
    const CopyDesc cd = descs[blockIdx.y];
    const int nx = cd.region.len(0), ny = cd.region.len(1);
    const long npts = cd.npts();
    const FabD s = src[cd.src_fab], d = dst[cd.dst_fab];
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        const int i = cd.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        const int j = cd.region.lo[1] + (int)(r % ny);
        const int k = cd.region.lo[2] + cd.kstep * (int)(r / ny);
        if (add) for (int n = 0; n < nc; ++n) d(i, j, k, dcomp + n) += s(i + cd.shift[0], j + cd.shift[1], k + cd.shift[2], scomp + n);
        else for (int n = 0; n < nc; ++n) d(i, j, k, dcomp + n) = s(i + cd.shift[0], j + cd.shift[1], k + cd.shift[2], scomp + n);
    }
}

void launch_copy_plan(const CopyDesc* d, int nd, long maxpts, const FabD* src, const FabD* dst, int scomp, int dcomp, int nc, hipStream_t s, bool add)
{
    if (nd == 0) return;
    // whole-array copies between layouts (one descriptor of millions of points) want thousands of workgroups in flight; a ghost-shell
    // exchange (many small descriptors: the grid's y dimension) keeps its short x dimension
    long nb = (maxpts + 255) / 256;
    const long cap = nd >= 8 ? 256 : 4096;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_copy_plan, dim3((unsigned)nb, (unsigned)nd), dim3(256), 0, s, d, src, dst, scomp, dcomp, nc, add ? 1 : 0);
}

__global__ void __launch_bounds__(256) k_pack(const CopyDesc* __restrict__ descs, const FabD* __restrict__ src,
                                              double* __restrict__ buf, long pts_total, int scomp, int nc)
{
    const CopyDesc cd = descs[blockIdx.y];
    const int nx = cd.region.len(0), ny = cd.region.len(1);
    const long npts = cd.npts();
    const FabD s = src[cd.src_fab];
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        const int i = cd.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        const int j = cd.region.lo[1] + (int)(r % ny);
        const int k = cd.region.lo[2] + cd.kstep * (int)(r / ny);
        for (int n = 0; n < nc; ++n) buf[cd.buf_off + q + pts_total * n] = s(i + cd.shift[0], j + cd.shift[1], k + cd.shift[2], scomp + n);
    }
}

__global__ void __launch_bounds__(256) k_unpack(const CopyDesc* __restrict__ descs, const FabD* __restrict__ dst,
                                                const double* __restrict__ buf, long pts_total, int dcomp, int nc, int add)
{
    const CopyDesc cd = descs[blockIdx.y];
    const int nx = cd.region.len(0), ny = cd.region.len(1);
    const long npts = cd.npts();
    const FabD d = dst[cd.dst_fab];
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        const int i = cd.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        const int j = cd.region.lo[1] + (int)(r % ny);
        const int k = cd.region.lo[2] + cd.kstep * (int)(r / ny);
        if (add) for (int n = 0; n < nc; ++n) d(i, j, k, dcomp + n) += buf[cd.buf_off + q + pts_total * n];
        else for (int n = 0; n < nc; ++n) d(i, j, k, dcomp + n) = buf[cd.buf_off + q + pts_total * n];
    }
}

void launch_pack(const CopyDesc* d, int nd, long maxpts, const FabD* src, double* buf, long pts_total, int scomp, int nc, hipStream_t s)
{
    if (nd == 0) return;
    long nb = (maxpts + 255) / 256; if (nb > 256) nb = 256; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)nb, (unsigned)nd), dim3(256), 0, s, d, src, buf, pts_total, scomp, nc);
}
void launch_unpack(const CopyDesc* d, int nd, long maxpts, const FabD* dst, const double* buf, long pts_total, int dcomp, int nc, hipStream_t s, bool add)
{
    if (nd == 0) return;
    long nb = (maxpts + 255) / 256; if (nb > 256) nb = 256; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_unpack, dim3((unsigned)nb, (unsigned)nd), dim3(256), 0, s, d, dst, buf, pts_total, dcomp, nc, add ? 1 : 0);
}

// ---- the same over a flat work list: workgroup b takes COPY_CHUNK points of descriptor work[b].x starting at work[b].y * COPY_CHUNK
template <int MODE>       // 0 copy, 1 pack, 2 unpack
__global__ void __launch_bounds__(256) k_copy_work(const CopyDesc* __restrict__ descs, const int2* __restrict__ work, const FabD* __restrict__ src,
                                                   const FabD* __restrict__ dst, double* __restrict__ buf, long pts_total, int scomp, int dcomp, int nc, int add)
{
    const int2 e = work[blockIdx.x];
    const CopyDesc cd = descs[e.x];
    const int nx = cd.region.len(0), ny = cd.region.len(1);
    const long npts = cd.npts();
    FabD s, d;
    if (MODE != 2) s = src[cd.src_fab];
    if (MODE != 1) d = dst[cd.dst_fab];
#pragma unroll
    for (int u = 0; u < COPY_CHUNK / 256; ++u) {
        const long q = (long)e.y * COPY_CHUNK + u * 256 + threadIdx.x;
        if (q >= npts) break;
        const int i = cd.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        const int j = cd.region.lo[1] + (int)(r % ny);
        const int k = cd.region.lo[2] + cd.kstep * (int)(r / ny);
        for (int n = 0; n < nc; ++n) {
            if (MODE == 0) {
                const double v = s(i + cd.shift[0], j + cd.shift[1], k + cd.shift[2], scomp + n);
                if (add) d(i, j, k, dcomp + n) += v; else d(i, j, k, dcomp + n) = v;
            } else if (MODE == 1) buf[cd.buf_off + q + pts_total * n] = s(i + cd.shift[0], j + cd.shift[1], k + cd.shift[2], scomp + n);
            else {
                const double v = buf[cd.buf_off + q + pts_total * n];
                if (add) d(i, j, k, dcomp + n) += v; else d(i, j, k, dcomp + n) = v;
            }
        }
    }
}
void launch_copy_plan_w(const CopyDesc* d, const int2* w, int nw, const FabD* src, const FabD* dst, int scomp, int dcomp, int nc, hipStream_t s, bool add)
{
    if (nw > 0) hipLaunchKernelGGL((k_copy_work<0>), dim3((unsigned)nw), dim3(256), 0, s, d, w, src, dst, (double*)nullptr, 0L, scomp, dcomp, nc, add ? 1 : 0);
}
void launch_pack_w(const CopyDesc* d, const int2* w, int nw, const FabD* src, double* buf, long pts_total, int scomp, int nc, hipStream_t s)
{
    if (nw > 0) hipLaunchKernelGGL((k_copy_work<1>), dim3((unsigned)nw), dim3(256), 0, s, d, w, src, (const FabD*)nullptr, buf, pts_total, scomp, 0, nc, 0);
}
void launch_unpack_w(const CopyDesc* d, const int2* w, int nw, const FabD* dst, const double* buf, long pts_total, int dcomp, int nc, hipStream_t s, bool add)
{
    if (nw > 0) hipLaunchKernelGGL((k_copy_work<2>), dim3((unsigned)nw), dim3(256), 0, s, d, w, (const FabD*)nullptr, dst, const_cast<double*>(buf), pts_total, 0, dcomp, nc, add ? 1 : 0);
}

// ------------------------------------------------------------------ reductions
// wavefront (64 lanes) shuffle reduction, then LDS across the 4 waves of the workgroup
template <int OP> __device__ __forceinline__ double red_op(double a, double b)
{
    if (OP == 0) return a + b;
    return a > b ? a : b;
}
template <int OP> __device__ __forceinline__ double block_reduce(double v)
{
    __shared__ double sm[4];
    for (int off = 32; off > 0; off >>= 1) v = red_op<OP>(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) v = red_op<OP>(red_op<OP>(sm[0], sm[1]), red_op<OP>(sm[2], sm[3]));
    return v;   // valid in thread 0
}

// stage 2: out[q] = reduce over partials[q*np .. q*np+np)
template <int OP> __global__ void __launch_bounds__(256) k_reduce_finish(const double* __restrict__ partials, int np, double* __restrict__ out)
{
    const int q = blockIdx.x;
    double v = OP == 2 ? -1.7976931348623157e308 : 0.0;       // OP 2: maximum of signed values
    for (int i = threadIdx.x; i < np; i += 256) v = red_op<OP>(v, partials[(size_t)q * np + i]);
    v = block_reduce<OP>(v);
    if (threadIdx.x == 0) out[q] = v;
}

__global__ void __launch_bounds__(256) k_norm0(Tiling t, const BoxD* __restrict__ boxes, int t0, int t1, int t2, int ng,
                                               const FabD* __restrict__ tab, int comp, int nc, double* __restrict__ partials)
{
    const int fab = tile_fab(t);
    const BoxD b = dev_grow_convert(boxes[fab], t0, t1, t2, ng);
    int i, j, k0, k1;
    double m = 0.0;
    if (tile_ijk(t, b, i, j, k0, k1)) {
        const FabD a = tab[fab];
        for (int n = 0; n < nc; ++n)
            for (int k = k0; k <= k1; ++k) { double v = fabs(a(i, j, k, comp + n)); m = v > m ? v : m; }
    }
    m = block_reduce<1>(m);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = m;
}

// Stage 2 of every reduction: the block partials -> nout values on the device; global: combined across the ranks IN PLACE on the device
// (Comm::allreduce_device: ncclAllReduce on the launch stream, no host round trip); then ONE read-back.  np == 0: this rank holds no
// data of the level and contributes the identity (0: sums, and maxima of absolute values).
static double finish_to_host(int op, int nout, int np, bool global)
{
    auto& ctx = Context::get();
    ctx.ensure_scratch((size_t)nout * np + 16);
    double* partials = ctx.d_scratch;
    double* out = ctx.d_scratch + (size_t)nout * np;
    if (np == 0 && op != 2) IAMRX_HIP_CHECK(hipMemsetAsync(out, 0, nout * sizeof(double), ctx.stream));
    else if (op == 0) hipLaunchKernelGGL((k_reduce_finish<0>), dim3(nout), dim3(256), 0, ctx.stream, partials, np, out);
    else if (op == 2) hipLaunchKernelGGL((k_reduce_finish<2>), dim3(nout), dim3(256), 0, ctx.stream, partials, np, out);
    else hipLaunchKernelGGL((k_reduce_finish<1>), dim3(nout), dim3(256), 0, ctx.stream, partials, np, out);
    if (global && ctx.comm->nranks > 1) ctx.comm->allreduce_device(out, nout, op == 0 ? ReduceOp::Sum : ReduceOp::Max, ctx.stream);
    IAMRX_HIP_CHECK(hipMemcpyAsync(ctx.h_scratch, out, nout * sizeof(double), hipMemcpyDeviceToHost, ctx.stream));
    ctx.sync();
    return ctx.h_scratch[0];
}

// the same, results left ON THE DEVICE at d_out[0 .. nout) (no read-back, no host synchronisation): the device-resident Krylov loop
// (krylov.h) consumes them from there.  op 0: sums, 1: maxima of absolute values
void reduce_finish_dev(int op, int nout, int np, bool global, double* d_out)
{
    auto& ctx = Context::get();
    ctx.ensure_scratch((size_t)nout * np + 16);
    double* partials = ctx.d_scratch;
    if (np == 0) IAMRX_HIP_CHECK(hipMemsetAsync(d_out, 0, nout * sizeof(double), ctx.stream));
    else if (op == 0) hipLaunchKernelGGL((k_reduce_finish<0>), dim3(nout), dim3(256), 0, ctx.stream, partials, np, d_out);
    else hipLaunchKernelGGL((k_reduce_finish<1>), dim3(nout), dim3(256), 0, ctx.stream, partials, np, d_out);
    if (global && ctx.comm->nranks > 1) ctx.comm->allreduce_device(d_out, nout, op == 0 ? ReduceOp::Sum : ReduceOp::Max, ctx.stream);
}

void reduce_finish_max(int nout, int np, bool global, double* out)
{
    auto& ctx = Context::get();
    if (np == 0 && !(global && ctx.comm->nranks > 1)) { for (int n = 0; n < nout; ++n) out[n] = 0.0; return; }
    finish_to_host(1, nout, np, global);
    for (int n = 0; n < nout; ++n) out[n] = ctx.h_scratch[n];
}

double reduce_norm0(const MultiFab& mf, int comp, int nc, int ng, bool global)
{
    global = global && !mf.layout->replicated;
    if (mf.nlocal() == 0) return (global && Context::get().comm->nranks > 1) ? finish_to_host(1, 1, 0, true) : 0.0;
    auto& ctx = Context::get();
    Tiling t = level_tiling(*mf.layout, mf.type, ng, 8, true);
    dim3 g = t.grid();
    const int np = (int)(g.x * g.y);
    ctx.ensure_scratch((size_t)np + 16);
    hipLaunchKernelGGL(k_norm0, g, Tiling::block(), 0, ctx.stream, t, mf.layout->d_boxes, mf.type.t[0], mf.type.t[1], mf.type.t[2], ng,
                       mf.d_tab, comp, nc, ctx.d_scratch);
    return finish_to_host(1, 1, np, global);
}

// largest and smallest value of one component in ONE pass and one read-back (partials: block maxima of v, then of -v)
__global__ void __launch_bounds__(256) k_minmax(Tiling t, const BoxD* __restrict__ boxes, int t0, int t1, int t2, int ng,
                                                const FabD* __restrict__ tab, int comp, double* __restrict__ partials, int np)
{
    const int fab = tile_fab(t);
    const BoxD b = dev_grow_convert(boxes[fab], t0, t1, t2, ng);
    int i, j, k0, k1;
    const bool in = tile_ijk(t, b, i, j, k0, k1);
    const FabD a = tab[fab];
    double hi = -1.7976931348623157e308, lo = -1.7976931348623157e308;
    if (in) for (int k = k0; k <= k1; ++k) { const double v = a(i, j, k, comp); hi = v > hi ? v : hi; lo = -v > lo ? -v : lo; }
    hi = block_reduce<1>(hi);
    lo = block_reduce<1>(lo);
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = hi;
        partials[(size_t)np + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = lo;
    }
}

// (a rank without data of the level contributes the identity -DBL_MAX to both: k_reduce_finish<2> over zero partials)
void reduce_minmax(const MultiFab& mf, int comp, int ng, double& mn, double& mx, bool global)
{
    global = global && !mf.layout->replicated;
    auto& ctx = Context::get();
    int np = 0;
    if (mf.nlocal() > 0) {
        Tiling t = level_tiling(*mf.layout, mf.type, ng, 8, true);
        dim3 g = t.grid();
        np = (int)(g.x * g.y);
        ctx.ensure_scratch((size_t)2 * np + 16);
        hipLaunchKernelGGL(k_minmax, g, Tiling::block(), 0, ctx.stream, t, mf.layout->d_boxes, mf.type.t[0], mf.type.t[1], mf.type.t[2], ng,
                           mf.d_tab, comp, ctx.d_scratch, np);
    }
    finish_to_host(2, 2, np, global);
    mx = ctx.h_scratch[0]; mn = -ctx.h_scratch[1];
}

// per-component maxima: partials[n * np + block]
__global__ void __launch_bounds__(256) k_norm0_comps(Tiling t, const BoxD* __restrict__ boxes, int t0, int t1, int t2, int ng,
                                                     const FabD* __restrict__ tab, int comp, int nc, double* __restrict__ partials, int np)
{
    const int fab = tile_fab(t);
    const BoxD b = dev_grow_convert(boxes[fab], t0, t1, t2, ng);
    int i, j, k0, k1;
    const bool in = tile_ijk(t, b, i, j, k0, k1);
    const FabD a = tab[fab];
    for (int n = 0; n < nc; ++n) {
        double m = 0.0;
        if (in) for (int k = k0; k <= k1; ++k) { double v = fabs(a(i, j, k, comp + n)); m = v > m ? v : m; }
        m = block_reduce<1>(m);
        if (threadIdx.x == 0) partials[(size_t)n * np + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
}

void reduce_norm0_comps(const MultiFab& mf, int comp, int nc, int ng, double* out, bool global)
{
    global = global && !mf.layout->replicated;
    auto& ctx = Context::get();
    int np = 0;
    if (mf.nlocal() > 0) {
        Tiling t = level_tiling(*mf.layout, mf.type, ng, 8, true);
        dim3 g = t.grid();
        np = (int)(g.x * g.y);
        ctx.ensure_scratch((size_t)nc * np + 16);
        hipLaunchKernelGGL(k_norm0_comps, g, Tiling::block(), 0, ctx.stream, t, mf.layout->d_boxes, mf.type.t[0], mf.type.t[1], mf.type.t[2], ng,
                           mf.d_tab, comp, nc, ctx.d_scratch, np);
    } else if (!(global && ctx.comm->nranks > 1)) { for (int n = 0; n < nc; ++n) out[n] = 0.0; return; }
    finish_to_host(1, nc, np, global);
    for (int n = 0; n < nc; ++n) out[n] = ctx.h_scratch[n];
}

// owner mask for nodal / face data: a box owns index hi+1 in a nodal direction only on a
// non-periodic domain boundary (elsewhere that point is the low point of a neighbouring box or a
// periodic image)
struct OwnerInfo { int type[3]; int dlo[3]; int dhi[3]; int per[3]; int half_lo[3]; int half_hi[3]; };
// weight of a point in sums / dot products: 0 for non-owner copies, 1/2 per Neumann wall a NODE lies on (the nodal
// system is stored in doubled form at wall nodes: MLNodeLinOp dot mask), 1 otherwise.
// Ownership rule: the hi+1 point of a box in a nodal direction belongs to the neighbouring box (or periodic image) as its low point,
// unless it lies on a non-periodic domain face.  On a level that does not cover the domain the hi+1 points on coarse/fine faces have
// no such neighbour and get weight 0: every caller on such levels (NodalMG with its Dirichlet mask, the composite solver's `own`
// masks, amrns.hip) holds zero / excluded values exactly there, so the sums are unaffected; face-centred sums are only taken on
// levels that cover the domain.
__device__ __forceinline__ double owner_weight(const OwnerInfo& o, const BoxD& cellbox, int i, int j, int k)
{
    const int idx[3] = {i, j, k};
    double w = 1.0;
    for (int d = 0; d < 3; ++d) {
        if (!o.type[d]) continue;
        if (idx[d] == cellbox.hi[d] + 1 && (o.per[d] || idx[d] != o.dhi[d] + 1)) return 0.0;
        if (o.half_lo[d] && idx[d] == o.dlo[d]) w *= 0.5;
        if (o.half_hi[d] && idx[d] == o.dhi[d] + 1) w *= 0.5;
    }
    return w;
}
static OwnerInfo make_owner(const MultiFab& m, const Geometry& g)
{
    OwnerInfo own;
    for (int d = 0; d < 3; ++d) {
        own.type[d] = m.type.t[d]; own.dlo[d] = g.domain.lo[d]; own.dhi[d] = g.domain.hi[d]; own.per[d] = g.periodic[d];
        own.half_lo[d] = g.half_lo[d]; own.half_hi[d] = g.half_hi[d];
    }
    return own;
}

__global__ void __launch_bounds__(256) k_dots(Tiling t, const BoxD* __restrict__ boxes, OwnerInfo own, int nout,
                                              const FabD* x0, const FabD* y0, const FabD* x1, const FabD* y1,
                                              int comp, int nc, double* __restrict__ partials, int np)
{
    const int fab = tile_fab(t);
    const BoxD cb = boxes[fab];
    const BoxD b = dev_grow_convert(cb, own.type[0], own.type[1], own.type[2], 0);
    int i, j, k0, k1;
    double s0 = 0.0, s1 = 0.0;
    if (tile_ijk(t, b, i, j, k0, k1)) {
        const FabD a0 = x0[fab], b0 = y0[fab];
        for (int n = 0; n < nc; ++n)
            for (int k = k0; k <= k1; ++k)
                { const double w = owner_weight(own, cb, i, j, k); if (w != 0.0) s0 += w * (a0(i, j, k, comp + n) * b0(i, j, k, comp + n)); }
        if (nout > 1) {
            const FabD a1 = x1[fab], b1 = y1[fab];
            for (int n = 0; n < nc; ++n)
                for (int k = k0; k <= k1; ++k)
                    { const double w = owner_weight(own, cb, i, j, k); if (w != 0.0) s1 += w * (a1(i, j, k, comp + n) * b1(i, j, k, comp + n)); }
        }
    }
    const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    s0 = block_reduce<0>(s0);
    if (threadIdx.x == 0) partials[slot] = s0;
    if (nout > 1) {
        s1 = block_reduce<0>(s1);
        if (threadIdx.x == 0) partials[(size_t)np + slot] = s1;
    }
}

void reduce_dots(int nout, const MultiFab* const* x, const MultiFab* const* y, int comp, int nc, const Geometry& g, double* out, bool local)
{
    IAMRX_ASSERT(nout == 1 || nout == 2);
    auto& ctx = Context::get();
    const MultiFab& m = *x[0];
    for (int q = 0; q < nout; ++q) out[q] = 0.0;
    const bool global = !local && !m.layout->replicated && ctx.comm->nranks > 1;
    if (m.nlocal() == 0 && global) { finish_to_host(0, nout, 0, true); for (int q = 0; q < nout; ++q) out[q] = ctx.h_scratch[q]; }
    if (m.nlocal() > 0) {
        Tiling t = level_tiling(*m.layout, m.type, 0, 8, true);
        dim3 gr = t.grid();
        const int np = (int)(gr.x * gr.y);
        ctx.ensure_scratch((size_t)np * nout + 16);
        const OwnerInfo own = make_owner(m, g);
        hipLaunchKernelGGL(k_dots, gr, Tiling::block(), 0, ctx.stream, t, m.layout->d_boxes, own, nout,
                           x[0]->d_tab, y[0]->d_tab, nout > 1 ? x[1]->d_tab : nullptr, nout > 1 ? y[1]->d_tab : nullptr,
                           comp, nc, ctx.d_scratch, np);
        finish_to_host(0, nout, np, global);
        for (int q = 0; q < nout; ++q) out[q] = ctx.h_scratch[q];
    }
}

// dot products left on the device (see reduce_finish_dev)
void reduce_dots_dev(int nout, const MultiFab* const* x, const MultiFab* const* y, int comp, int nc, const Geometry& g, double* d_out, bool local)
{
    IAMRX_ASSERT(nout == 1 || nout == 2);
    auto& ctx = Context::get();
    const MultiFab& m = *x[0];
    const bool global = !local && !m.layout->replicated && ctx.comm->nranks > 1;
    int np = 0;
    if (m.nlocal() > 0) {
        Tiling t = level_tiling(*m.layout, m.type, 0, 8, true);
        dim3 gr = t.grid();
        np = (int)(gr.x * gr.y);
        ctx.ensure_scratch((size_t)np * nout + 16);
        const OwnerInfo own = make_owner(m, g);
        hipLaunchKernelGGL(k_dots, gr, Tiling::block(), 0, ctx.stream, t, m.layout->d_boxes, own, nout,
                           x[0]->d_tab, y[0]->d_tab, nout > 1 ? x[1]->d_tab : nullptr, nout > 1 ? y[1]->d_tab : nullptr,
                           comp, nc, ctx.d_scratch, np);
    }
    reduce_finish_dev(0, nout, np, global, d_out);
}

__global__ void __launch_bounds__(256) k_sum_unique(Tiling t, const BoxD* __restrict__ boxes, OwnerInfo own,
                                                    const FabD* __restrict__ tab, int comp, double* __restrict__ partials)
{
    const int fab = tile_fab(t);
    const BoxD cb = boxes[fab];
    const BoxD b = dev_grow_convert(cb, own.type[0], own.type[1], own.type[2], 0);
    int i, j, k0, k1;
    double s = 0.0;
    if (tile_ijk(t, b, i, j, k0, k1)) {
        const FabD a = tab[fab];
        for (int k = k0; k <= k1; ++k)
            { const double w = owner_weight(own, cb, i, j, k); if (w != 0.0) s += w * a(i, j, k, comp); }
    }
    s = block_reduce<0>(s);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

double reduce_sum_unique(const MultiFab& mf, int comp, const Geometry& g, bool global)
{
    global = global && !mf.layout->replicated;
    if (mf.nlocal() == 0) return (global && Context::get().comm->nranks > 1) ? finish_to_host(0, 1, 0, true) : 0.0;
    auto& ctx = Context::get();
    Tiling t = level_tiling(*mf.layout, mf.type, 0, 8, true);
    dim3 gr = t.grid();
    const int np = (int)(gr.x * gr.y);
    ctx.ensure_scratch((size_t)np + 16);
    const OwnerInfo own = make_owner(mf, g);
    hipLaunchKernelGGL(k_sum_unique, gr, Tiling::block(), 0, ctx.stream, t, mf.layout->d_boxes, own, mf.d_tab, comp, ctx.d_scratch);
    return finish_to_host(0, 1, np, global);
}

// ------------------------------------------------------------------ BLAS-1 style
void mf_lincomb(MultiFab& dst, double a, const MultiFab& x, double b, const MultiFab& y, int comp, int nc, int ng)
{
    trace_blas_site("lincomb", dst.layout ? dst.layout->local_cells() * nc : 0);
    if (!dst.base) return;
    const FabD *dt = dst.d_tab, *xt = x.d_tab, *yt = y.d_tab;
    for_each_2ph(*dst.layout, dst.type, ng, nc, Context::get().stream,
        [=] __device__(int i, int j, int k, int f, int n) { return a * xt[f](i, j, k, comp + n) + b * yt[f](i, j, k, comp + n); },
        [=] __device__(int i, int j, int k, int f, int n, double v) { dt[f](i, j, k, comp + n) = v; });
}

void mf_saxpy(MultiFab& y, double a, const MultiFab& x, int xcomp, int ycomp, int nc, int ng)
{
    trace_blas_site("saxpy", y.layout ? y.layout->local_cells() * nc : 0);
    if (!y.base) return;
    const FabD *yt = y.d_tab, *xt = x.d_tab;
    for_each_2ph(*y.layout, y.type, ng, nc, Context::get().stream,
        [=] __device__(int i, int j, int k, int f, int n) { return yt[f](i, j, k, ycomp + n) + a * xt[f](i, j, k, xcomp + n); },
        [=] __device__(int i, int j, int k, int f, int n, double v) { yt[f](i, j, k, ycomp + n) = v; });
}

void mf_add_scalar(MultiFab& y, double a, int comp, int nc, int ng)
{
    if (!y.base) return;
    const FabD* yt = y.d_tab;
    for_each(*y.layout, y.type, ng, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD yy = yt[f];
        for (int n = 0; n < nc; ++n) yy(i, j, k, comp + n) += a;
    });
}

void mf_mult(MultiFab& y, double a, int comp, int nc, int ng)
{
    trace_blas_site("mult", y.layout ? y.layout->local_cells() * nc : 0);
    if (!y.base) return;
    const FabD* yt = y.d_tab;
    for_each_2ph(*y.layout, y.type, ng, nc, Context::get().stream,
        [=] __device__(int i, int j, int k, int f, int n) { return yt[f](i, j, k, comp + n) * a; },
        [=] __device__(int i, int j, int k, int f, int n, double v) { yt[f](i, j, k, comp + n) = v; });
}

// slab levels (mf.h): every y-plane of dst takes the single y-plane of src
void slab_duplicate(MultiFab& dst, const MultiFab& src)
{
    IAMRX_ASSERT(dst.ncomp == src.ncomp && dst.layout->boxes.size() == src.layout->boxes.size());
    if (dst.nlocal() == 0) return;
    const FabD *dt = dst.d_tab, *st = src.d_tab;
    const int nc = dst.ncomp;
    for_each(*dst.layout, dst.type, 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD s = st[f];
        for (int n = 0; n < nc; ++n) dt[f](i, j, k, n) = s(i, 0, k, n);        // (slab boxes start at y = 0: Layout::slab_coarsenable)
    });
}

}  // namespace iamrx
