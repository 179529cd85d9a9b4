// iamr_amd/csrc/k_nodal.hip -- nodal (Q1 finite element, cell-centred sigma) Laplacian kernels for gfx950:
// 27-point operator / residual, 8-colour Gauss-Seidel, weighted Jacobi, full-weighting restriction,
// sigma-weighted interpolation, nodal divergence, velocity update and cell-centred gradient.
//
// Role: AMReX MLNodeLaplacian (mlndlap_adotx_aa, gscolor_aa / jacobi_aa, restriction, interpadd_aa, divu,
// mknewu_aa) driven by Hydro::NodalProjector at reference Source/Projection.cpp:2512-2542, and
// MLNodeLaplacian::compGrad at Source/NavierStokesBase.cpp:4106-4118.
//
// The 27 stencil coefficients are written per neighbour class (corner / edge / face / centre): the
// coefficient of a neighbour is w(class) * (sum of sigma over the cells shared with that neighbour).
#include "kernels.h"
#include <cstring>
#include <type_traits>
#include "launch.h"

namespace iamrx {

struct NodeW { double corner, ex, ey, ez, fx, fy, fz, c; };   // class weights

static NodeW make_w(const Geometry& g)
{
    const double fx = 1.0 / (g.dx[0] * g.dx[0]) / 36.0, fy = 1.0 / (g.dx[1] * g.dx[1]) / 36.0, fz = 1.0 / (g.dx[2] * g.dx[2]) / 36.0;
    NodeW w;
    w.corner = fx + fy + fz;
    w.ex = -fx + 2.0 * fy + 2.0 * fz;     // edge neighbours with di = 0
    w.ey = 2.0 * fx - fy + 2.0 * fz;      // dj = 0
    w.ez = 2.0 * fx + 2.0 * fy - fz;      // dk = 0
    w.fx = 4.0 * fx - 2.0 * fy - 2.0 * fz;   // face neighbours (di = +-1, dj = dk = 0)
    w.fy = -2.0 * fx + 4.0 * fy - 2.0 * fz;
    w.fz = -2.0 * fx - 2.0 * fy + 4.0 * fz;
    w.c = -4.0 * (fx + fy + fz);
    return w;
}

// A x at node (i,j,k) and the diagonal coefficient s0.  s = sigma fab (cell centred, 1 ghost).
__device__ __forceinline__ double node_Ax(const FabD& x, const FabD& s, const NodeW& w, int i, int j, int k, double& s0)
{
    // sigma of the 8 cells around the node: s_{abc}, a,b,c in {m,p} for cell index node-1 / node
    const double smmm = s(i - 1, j - 1, k - 1), spmm = s(i, j - 1, k - 1), smpm = s(i - 1, j, k - 1), sppm = s(i, j, k - 1);
    const double smmp = s(i - 1, j - 1, k), spmp = s(i, j - 1, k), smpp = s(i - 1, j, k), sppp = s(i, j, k);
    s0 = w.c * (smmm + spmm + smpm + sppm + smmp + spmp + smpp + sppp);
    double y = x(i, j, k) * s0;
    y += w.corner * (x(i - 1, j - 1, k - 1) * smmm + x(i + 1, j - 1, k - 1) * spmm + x(i - 1, j + 1, k - 1) * smpm + x(i + 1, j + 1, k - 1) * sppm
                   + x(i - 1, j - 1, k + 1) * smmp + x(i + 1, j - 1, k + 1) * spmp + x(i - 1, j + 1, k + 1) * smpp + x(i + 1, j + 1, k + 1) * sppp);
    y += w.ex * (x(i, j - 1, k - 1) * (smmm + spmm) + x(i, j + 1, k - 1) * (smpm + sppm) + x(i, j - 1, k + 1) * (smmp + spmp) + x(i, j + 1, k + 1) * (smpp + sppp));
    y += w.ey * (x(i - 1, j, k - 1) * (smmm + smpm) + x(i + 1, j, k - 1) * (spmm + sppm) + x(i - 1, j, k + 1) * (smmp + smpp) + x(i + 1, j, k + 1) * (spmp + sppp));
    y += w.ez * (x(i - 1, j - 1, k) * (smmm + smmp) + x(i + 1, j - 1, k) * (spmm + spmp) + x(i - 1, j + 1, k) * (smpm + smpp) + x(i + 1, j + 1, k) * (sppm + sppp));
    y += w.fx * (x(i - 1, j, k) * (smmm + smpm + smmp + smpp) + x(i + 1, j, k) * (spmm + sppm + spmp + sppp));
    y += w.fy * (x(i, j - 1, k) * (smmm + spmm + smmp + spmp) + x(i, j + 1, k) * (smpm + sppm + smpp + sppp));
    y += w.fz * (x(i, j, k - 1) * (smmm + spmm + smpm + sppm) + x(i, j, k + 1) * (smmp + spmp + smpp + sppp));
    return y;
}

// out = rhs - A x   (rhs null: out = A x).  z-marching: a workgroup owns a TXxTY column of nodes and walks KC planes; the
// three x-planes and two sigma-planes it needs live in LDS (rolling), every plane is fetched from HBM once per column
// (+ the 1-node halo ring), and the loads of plane k+2 are in flight while plane k is being evaluated.
void nodal_residual_launch(const Geometry& g, MultiFab& out, const MultiFab& x, const MultiFab& sig, const MultiFab* rhs, unsigned long long* d_norm);
// normout != null: the launch also reduces the max norm of what it writes (wave maximum, one atomicMax per wavefront on the bit pattern of
// the non-negative value, NaN -> +inf: order independent, hence deterministic; k_abec.hip's norm_commit)
__device__ __forceinline__ void nodal_norm_commit(double mx, unsigned long long* out)
{
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
        if (bits > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, bits);
    }
}
template <int TX, int TY>
__global__ void __launch_bounds__(TX * TY) k_nodal_residual_zm(const BoxD* __restrict__ boxes, const FabD* __restrict__ ot, const FabD* __restrict__ xt,
    const FabD* __restrict__ st, const FabD* __restrict__ rt, NodeW w, int ntx, int nty, int kc, unsigned long long* __restrict__ normout,
    const int4* __restrict__ list)
{
    double mx = 0.0;
    constexpr int RX = TX + 2, RY = TY + 2, NT = TX * TY, NLD = (RX * RY + NT - 1) / NT;
    __shared__ double X[3][RY][RX];
    __shared__ double S[2][RY][RX];
    // list (a level of many unequal boxes): entry b = (box, tile x, tile y, z-chunk) of the tiles that exist, instead of the tile grid of the
    // largest box for every box
    int fab = blockIdx.y, tix, tiy, ck;
    if (list) { const int4 e = list[blockIdx.x]; fab = e.x; tix = e.y; tiy = e.z; ck = e.w; }
    else { const int bid = blockIdx.x; tix = bid % ntx; const int r1 = bid / ntx; tiy = r1 % nty; ck = r1 / nty; }
    const BoxD cb = boxes[fab];
    const int nhi0 = cb.hi[0] + 1, nhi1 = cb.hi[1] + 1, nhi2 = cb.hi[2] + 1;
    const int tx0 = cb.lo[0] + tix * TX, ty0 = cb.lo[1] + tiy * TY, k0 = cb.lo[2] + ck * kc;
    if (tx0 > nhi0 || ty0 > nhi1 || k0 > nhi2) return;
    const int k1 = min(k0 + kc - 1, nhi2);
    const FabD x = xt[fab], s = st[fab], o = ot[fab];
    const int ox = tx0 - 1, oy = ty0 - 1;
    const int tid = threadIdx.x;
    // footprint addressing (clamped into the arrays; clamped entries are never used by a valid node)
    long xoff[NLD], soff[NLD];
    int lidx[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int idx = min(tid + it * NT, RX * RY - 1);
        lidx[it] = idx;
        const int gi = ox + idx % RX, gj = oy + idx / RX;
        const int xi = min(max(gi, x.lo[0]), x.lo[0] + x.n[0] - 1), xj = min(max(gj, x.lo[1]), x.lo[1] + x.n[1] - 1);
        const int si = min(max(gi, s.lo[0]), s.lo[0] + s.n[0] - 1), sj = min(max(gj, s.lo[1]), s.lo[1] + s.n[1] - 1);
        xoff[it] = x.off(xi, xj, x.lo[2]);
        soff[it] = s.off(si, sj, s.lo[2]);
    }
    const long xpl = (long)x.n[0] * x.n[1], spl = (long)s.n[0] * s.n[1];
    const FabD::gdouble* xp = (const FabD::gdouble*)x.p;
    const FabD::gdouble* sp = (const FabD::gdouble*)s.p;
    auto ldx = [&](int it, int k) { return xp[xoff[it] + xpl * (k - x.lo[2])]; };
    auto lds_ = [&](int it, int k) { return sp[soff[it] + spl * (k - s.lo[2])]; };
    // prologue: planes k0-1, k0 of x and cell plane k0-1 of sigma
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int ly = lidx[it] / RX, lx = lidx[it] % RX;
        X[(k0 + 2) % 3][ly][lx] = ldx(it, k0 - 1);
        X[k0 % 3][ly][lx] = ldx(it, k0);
        S[(k0 + 1) & 1][ly][lx] = lds_(it, k0 - 1);
    }
    double vx[NLD], vs[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) { vx[it] = ldx(it, k0 + 1); vs[it] = lds_(it, k0); }
    const int lx = tid % TX + 1, ly = tid / TX + 1;
    const int i = ox + lx, j = oy + ly;
    const bool valid = i <= nhi0 && j <= nhi1;
    for (int k = k0; k <= k1; ++k) {
        const int a0 = (k + 2) % 3, a1 = k % 3, a2 = (k + 1) % 3, b0 = (k + 1) & 1, b1 = k & 1;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int qy = lidx[it] / RX, qx = lidx[it] % RX;
            X[a2][qy][qx] = vx[it];
            S[b1][qy][qx] = vs[it];
        }
        __syncthreads();
        if (k < k1) {
#pragma unroll
            for (int it = 0; it < NLD; ++it) { vx[it] = ldx(it, k + 2); vs[it] = lds_(it, k + 1); }
        }
        if (valid) {
            const double smmm = S[b0][ly - 1][lx - 1], spmm = S[b0][ly - 1][lx], smpm = S[b0][ly][lx - 1], sppm = S[b0][ly][lx];
            const double smmp = S[b1][ly - 1][lx - 1], spmp = S[b1][ly - 1][lx], smpp = S[b1][ly][lx - 1], sppp = S[b1][ly][lx];
            const double s0 = w.c * (smmm + spmm + smpm + sppm + smmp + spmp + smpp + sppp);
            double y = X[a1][ly][lx] * s0;
            y += w.corner * (X[a0][ly - 1][lx - 1] * smmm + X[a0][ly - 1][lx + 1] * spmm + X[a0][ly + 1][lx - 1] * smpm + X[a0][ly + 1][lx + 1] * sppm
                           + X[a2][ly - 1][lx - 1] * smmp + X[a2][ly - 1][lx + 1] * spmp + X[a2][ly + 1][lx - 1] * smpp + X[a2][ly + 1][lx + 1] * sppp);
            y += w.ex * (X[a0][ly - 1][lx] * (smmm + spmm) + X[a0][ly + 1][lx] * (smpm + sppm) + X[a2][ly - 1][lx] * (smmp + spmp) + X[a2][ly + 1][lx] * (smpp + sppp));
            y += w.ey * (X[a0][ly][lx - 1] * (smmm + smpm) + X[a0][ly][lx + 1] * (spmm + sppm) + X[a2][ly][lx - 1] * (smmp + smpp) + X[a2][ly][lx + 1] * (spmp + sppp));
            y += w.ez * (X[a1][ly - 1][lx - 1] * (smmm + smmp) + X[a1][ly - 1][lx + 1] * (spmm + spmp) + X[a1][ly + 1][lx - 1] * (smpm + smpp) + X[a1][ly + 1][lx + 1] * (sppm + sppp));
            y += w.fx * (X[a1][ly][lx - 1] * (smmm + smpm + smmp + smpp) + X[a1][ly][lx + 1] * (spmm + sppm + spmp + sppp));
            y += w.fy * (X[a1][ly - 1][lx] * (smmm + spmm + smmp + spmp) + X[a1][ly + 1][lx] * (smpm + sppm + smpp + sppp));
            y += w.fz * (X[a0][ly][lx] * (smmm + spmm + smpm + sppm) + X[a2][ly][lx] * (smmp + spmp + smpp + sppp));
            const double v = rt ? rt[fab](i, j, k) - y : y;
            o(i, j, k) = v;
            const double a = fabs(v);
            mx = fmax(mx, a == a ? a : INFINITY);
        }
        __syncthreads();
    }
    if (normout) nodal_norm_commit(mx, normout);
}

// norm_out != null: *norm_out = max norm of out over all ranks if the z-marching kernel ran (returns true); false: the caller computes it
bool nodal_residual(const Geometry& g, MultiFab& out, const MultiFab& x, const MultiFab& sig, const MultiFab* rhs, double* norm_out)
{
    auto& ctx = Context::get();
    static unsigned long long* d_norm = nullptr;
    if (norm_out && !d_norm) IAMRX_HIP_CHECK(hipMalloc(&d_norm, 2 * sizeof(unsigned long long)));
    const Layout& lay = *x.layout;
    const bool zm = tune("NODAL_RES_ZM", 1) != 0 && lay.max_len[0] >= 16 && lay.max_len[1] >= 8 && x.ngrow >= 1 && sig.ngrow >= 1;
    const bool fused = norm_out && zm && tune("NODAL_RES_NORM", 1) != 0;
    if (fused) IAMRX_HIP_CHECK(hipMemsetAsync(d_norm, 0, sizeof(unsigned long long), ctx.stream));
    nodal_residual_launch(g, out, x, sig, rhs, fused ? d_norm : nullptr);
    if (!fused) return false;
    const bool global = !lay.replicated && ctx.comm->nranks > 1;
    if (global) ctx.comm->allreduce_device(reinterpret_cast<double*>(d_norm), 1, ReduceOp::Max, ctx.stream);
    unsigned long long bits = 0;
    IAMRX_HIP_CHECK(hipMemcpyAsync(&bits, d_norm, sizeof(bits), hipMemcpyDeviceToHost, ctx.stream));
    ctx.sync();
    double v;
    std::memcpy(&v, &bits, sizeof(v));
    *norm_out = v;
    return true;
}

void nodal_residual_launch(const Geometry& g, MultiFab& out, const MultiFab& x, const MultiFab& sig, const MultiFab* rhs, unsigned long long* d_norm)
{
    if (x.nlocal() == 0) return;
    const NodeW w = make_w(g);
    const FabD *ot = out.d_tab, *xt = x.d_tab, *st = sig.d_tab;
    const FabD* rt = rhs ? rhs->d_tab : nullptr;
    const Layout& l = *x.layout;
    const bool zmarch = tune("NODAL_RES_ZM", 1) != 0;
    if (zmarch && l.max_len[0] >= 16 && l.max_len[1] >= 8 && x.ngrow >= 1 && sig.ngrow >= 1) {
        constexpr int TX = 32, TY = 8;
        const int ntx = (l.max_len[0] + 1 + TX - 1) / TX, nty = (l.max_len[1] + 1 + TY - 1) / TY;
        const int nk = l.max_len[2] + 1;
        const int kc = nk >= 128 ? 32 : (nk >= 32 ? 16 : nk);
        const int nck = (nk + kc - 1) / kc;
        if (l.nlocal() >= 4 && tune("TILE_LISTS", 1) != 0) {
            int n = 0;
            const int4* lst = layout_int4_list(l, {3, TX, TY, kc, 0}, [&](std::vector<int4>& h) {
                for (int f = 0; f < l.nlocal(); ++f) {
                    const BoxD b = l.lbox(f);
                    const int bx = (b.len(0) + 1 + TX - 1) / TX, by = (b.len(1) + 1 + TY - 1) / TY, bk = (b.len(2) + 1 + kc - 1) / kc;
                    for (int c = 0; c < bk; ++c) for (int ty = 0; ty < by; ++ty) for (int tx = 0; tx < bx; ++tx) h.push_back(make_int4(f, tx, ty, c));
                }
            }, &n);
            if (lst && 4L * n <= 3L * ntx * nty * nck * l.nlocal()) {
                hipLaunchKernelGGL((k_nodal_residual_zm<TX, TY>), dim3((unsigned)n), dim3(TX * TY), 0, Context::get().stream, l.d_boxes, ot, xt, st, rt, w, ntx, nty, kc,
                                   d_norm, lst);
                return;
            }
        }
        dim3 grid((unsigned)(ntx * nty * nck), (unsigned)l.nlocal());
        hipLaunchKernelGGL((k_nodal_residual_zm<TX, TY>), grid, dim3(TX * TY), 0, Context::get().stream, l.d_boxes, ot, xt, st, rt, w, ntx, nty, kc, d_norm,
                           (const int4*)nullptr);
        return;
    }
    for_each(*x.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        double s0;
        const double y = node_Ax(xt[f], st[f], w, i, j, k, s0);
        ot[f](i, j, k) = rt ? rt[f](i, j, k) - y : y;
    });
}

// one colour of the 8-colour Gauss-Seidel sweep: nodes with (i&1, j&1, k&1) == colour bits
__global__ void __launch_bounds__(256) k_nodal_gscolor(Tiling t, const BoxD* __restrict__ boxes, const FabD* __restrict__ xt,
    const FabD* __restrict__ rt, const FabD* __restrict__ st, NodeW w, int color, const FabD* __restrict__ dmt)
{
    const int fab = blockIdx.y;
    const BoxD cb = boxes[fab];
    // nodal valid box [lo, hi+1]; first index of the right parity in each direction
    const int cx = color & 1, cy = (color >> 1) & 1, cz = (color >> 2) & 1;
    const int i0 = cb.lo[0] + (((cb.lo[0] & 1) != cx) ? 1 : 0);
    const int j0 = cb.lo[1] + (((cb.lo[1] & 1) != cy) ? 1 : 0);
    const int k0n = cb.lo[2] + (((cb.lo[2] & 1) != cz) ? 1 : 0);
    BoxD hb;   // index box of the colour lattice (m-space)
    hb.lo[0] = hb.lo[1] = hb.lo[2] = 0;
    hb.hi[0] = (cb.hi[0] + 1 - i0) >> 1; hb.hi[1] = (cb.hi[1] + 1 - j0) >> 1; hb.hi[2] = (cb.hi[2] + 1 - k0n) >> 1;
    if (cb.hi[0] + 1 < i0 || cb.hi[1] + 1 < j0 || cb.hi[2] + 1 < k0n) return;
    int mi, mj, mk0, mk1;
    if (!tile_ijk(t, hb, mi, mj, mk0, mk1)) return;
    const FabD x = xt[fab], r = rt[fab], s = st[fab];
    const int i = i0 + 2 * mi, j = j0 + 2 * mj;
    for (int mk = mk0; mk <= mk1; ++mk) {
        const int k = k0n + 2 * mk;
        if (dmt && dmt[fab](i, j, k) != 0.0) continue;      // Dirichlet node: keeps its value
        double s0;
        const double Ax = node_Ax(x, s, w, i, j, k, s0);
        x(i, j, k) += (r(i, j, k) - Ax) / s0;
    }
}

void nodal_gs_color(const Geometry& g, MultiFab& x, const MultiFab& rhs, const MultiFab& sig, int color, const MultiFab* dmask)
{
    if (x.nlocal() == 0) return;
    const Layout& l = *x.layout;
    int ml[3];
    for (int d = 0; d < 3; ++d) ml[d] = (l.max_len[d] + 1 + 1) / 2;
    Tiling t = make_tiling(ml, l.nlocal(), 4);
    hipLaunchKernelGGL(k_nodal_gscolor, t.grid(), Tiling::block(), 0, Context::get().stream, t, l.d_boxes, x.d_tab, rhs.d_tab, sig.d_tab, make_w(g), color,
                       dmask ? dmask->d_tab : nullptr);
}

// ------------------------------------------------------------------------------------------------------
// Plane-fused 8-colour Gauss-Seidel: the four colours of one k-parity (colours 0-3: k even, 4-7: k odd)
// are applied in ONE pass.  A workgroup stages the x-planes k-1,k,k+1 and the two sigma cell planes of a
// (TX+8)x(TY+8) footprint in LDS, runs the four in-plane colour updates back to back on shrinking regions
// (grown by 3,2,1,0 nodes: the halo is recomputed instead of exchanged) and writes the interior TXxTY nodes
// of plane k.  The k+-1 planes belong to the other parity and are read-only during the pass, so the result
// is identical to four sequential colour passes with a ghost fill in front of each of them -- provided the
// arrays carry 4 ghost nodes (x, sigma) / 3 (rhs) that hold true periodic / neighbour images.
// The pass is OUT OF PLACE: plane k is read from xc (state before the pass), the k+-1 planes from xn and the result goes
// to xo.  In place, a workgroup could read halo nodes of plane k that a neighbouring workgroup has already updated (not
// all workgroups of a large level are resident together), which would silently change the Gauss-Seidel ordering.
// HBM traffic per sweep drops from 8 full-array passes to 2; kernel launches from 8+8 fills to 2+2.
// periodic image of a node / cell index inside the one box [lo, hi] (cells) that spans the periodic domain
__device__ __forceinline__ int wrap_node(int g, int lo, int hi)
{
    // the staging halo is 4 nodes wide and the box has >= 4 cells (periodic_wrap_ok(g, l, 4)): one conditional shift is enough
    const int n = hi - lo + 1;
    return g < lo ? g + n : (g > hi + 1 ? g - n : g);
}
__device__ __forceinline__ int wrap_cell(int g, int lo, int hi)
{
    const int n = hi - lo + 1;
    return g < lo ? g + n : (g > hi ? g - n : g);
}
// ... or, in a direction that ends on Neumann walls (refl): the mirror image -- what nodal_reflect_bc / cc_mirror_bc write into the ghost
// nodes / cells (even reflection about the wall node lo resp. hi + 1; sigma mirrored about the wall face): the index-wrap variants then
// read the valid data of a wall-bounded box directly as well, no ghost fill in front of a pass
__device__ __forceinline__ int image_node(int g, int lo, int hi, bool refl)
{
    if (refl) return g < lo ? 2 * lo - g : (g > hi + 1 ? 2 * (hi + 1) - g : g);
    return wrap_node(g, lo, hi);
}
__device__ __forceinline__ int image_cell(int g, int lo, int hi, bool refl)
{
    if (refl) return g < lo ? 2 * lo - 1 - g : (g > hi ? 2 * hi + 1 - g : g);
    return wrap_cell(g, lo, hi);
}

template <int TX, int TY, int NT, bool WRAP, bool MASK, bool CSIG>
// no minimum-occupancy bound: measured at 256^3 on MI355X (profiles/round2_b_*), forcing 4 waves/SIMD on the variable-sigma variant (168 VGPRs
// -> 128 + 41 spilled) costs 2x (325 us against 165 us per launch), 5-6 on the constant-sigma one (104 VGPRs) gains nothing / loses 35 %
// the masked variant (levels with Dirichlet nodes: refined AMR levels, outflow faces) needs 181 VGPRs unconstrained -- 2 wavefronts per SIMD,
// 230 us per 257^3 launch against 161 us for the unmasked one at 168 VGPRs / 3 wavefronts: it is held to 3 (a handful of spills)
__global__ void __launch_bounds__(NT, (MASK && NT == 256 ? 3 : 1)) k_nodal_gs4(const BoxD* __restrict__ boxes, const FabD* __restrict__ xct, const FabD* __restrict__ xnt,
    const FabD* __restrict__ xot, const FabD* __restrict__ rt, const FabD* __restrict__ st, NodeW w, int kpar, int ntx, int nty, int xcd_chunk,
    const FabD* __restrict__ dmt, double csig, int ppc, int refl)
{
    // A workgroup owns one TXxTY tile and marches through ppc consecutive planes of its parity (k, k+2, ...).  Plane k+1 staged
    // for plane k is the k-1 plane of the next one and stays in LDS (two x planes loaded per plane instead of three); the loads of
    // the next plane are issued BEFORE the four colour passes of the current one and land in registers while those run, so the
    // global-load latency that bounded the one-plane-per-workgroup version is hidden behind the LDS phase.
    // LDS rows are stored parity-split: column lx lives at (lx&1)*HX + (lx>>1).  A colour pass touches every second
    // column, so its 64 lanes then read consecutive doubles (no bank conflicts) instead of a stride-2 pattern.
    constexpr int RX = TX + 8, RY = TY + 8, HX = RX / 2, PX = RX + 2;   // PX: padded row pitch
    constexpr int PL = RY * PX;                                         // one staged plane
    __shared__ double Xf[3 * PL];
    // CSIG: sigma is one constant on the whole level (constant-density flow): no sigma planes in LDS (24 KB per workgroup: 6 per CU)
    __shared__ double Sf[CSIG ? 1 : 2 * PL];
    const int fab = blockIdx.y;
    const BoxD cb = boxes[fab];
    int tix, tiy, pk;
    if (xcd_chunk > 0) {
        // XCD-aware order (workgroup b runs on XCD b % 8, speed only): every XCD sweeps its own contiguous slab of tiles chunk
        // by chunk, so the halo overlap of neighbouring tiles stays in its L2
        const int q = blockIdx.x & 7, m = blockIdx.x >> 3;
        const int nt = ntx * nty;
        const int tlo = (q * nt) >> 3, cnt = (((q + 1) * nt) >> 3) - tlo;
        if (cnt == 0 || m >= cnt * xcd_chunk) return;
        const int t = tlo + m % cnt;
        pk = m / cnt;
        tix = t % ntx; tiy = t / ntx;
    } else {
        const int bid = blockIdx.x;
        tix = bid % ntx;
        const int r1 = bid / ntx;
        tiy = r1 % nty; pk = r1 / nty;
    }
    const int kfirst = cb.lo[2] + (((cb.lo[2] & 1) != kpar) ? 1 : 0);
    const int k0 = kfirst + 2 * pk * ppc;
    const int nhi0 = cb.hi[0] + 1, nhi1 = cb.hi[1] + 1, nhi2 = cb.hi[2] + 1;   // last valid node
    const int tx0 = cb.lo[0] + tix * TX, ty0 = cb.lo[1] + tiy * TY;
    if (k0 > nhi2 || tx0 > nhi0 || ty0 > nhi1) return;
    const int kend = min(k0 + 2 * (ppc - 1), nhi2);                            // last plane of the chunk is the last k <= kend of the parity
    const int txe = min(tx0 + TX - 1, nhi0), tye = min(ty0 + TY - 1, nhi1);
    const FabD x = xct[fab], xn = xnt[fab], xo = xot[fab], r = rt[fab], s = st[fab];
    const int ox = tx0 - 4, oy = ty0 - 4;
    const int tid = threadIdx.x;
#define COL(lx) ((((lx) & 1) * HX) + ((lx) >> 1))
    // colour pass c updates the tile grown by 3 - c nodes; every thread owns at most one node of a pass
    static_assert(((TX + 7) / 2) * ((TY + 7) / 2) <= NT, "one node per thread and colour pass");
    int q0[4], qm[4];     // LDS offsets of the node of this thread in pass c (row * PX + column) and of its lx-1 neighbour (lx+1 sits at qm + 1); q0 < 0: none
    int roff[4], doff[4]; // offsets of that node in plane 0 of the right-hand side / the Dirichlet mask
    double rr[4];
    bool fixed[4];        // MASK: the node of this thread in pass c is a Dirichlet node (keeps its value)
    // A node strictly inside the box has its eight cells in the box: it is never a Dirichlet node (nodal_build_dmask marks nodes with a
    // cell outside the level, and domain faces).  The mask is read only for nodes on the box surface or in the ghost region -- a few
    // per cent of them -- instead of 8 bytes for every node of every pass.
    bool surf[4];
    const int rks = r.n[0] * r.n[1];
    int dks = 0;
    if constexpr (MASK) dks = dmt[fab].n[0] * dmt[fab].n[1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cx = c & 1, cy = c >> 1, g = 3 - c;
        const int rlo0 = tx0 - g, rhi0 = txe + g, rlo1 = ty0 - g, rhi1 = tye + g;
        const int i0 = rlo0 + (((rlo0 & 1) != cx) ? 1 : 0), j0 = rlo1 + (((rlo1 & 1) != cy) ? 1 : 0);
        const int ni = i0 > rhi0 ? 0 : ((rhi0 - i0) >> 1) + 1, nj = j0 > rhi1 ? 0 : ((rhi1 - j0) >> 1) + 1;
        const bool on = tid < ni * nj;
        const int pi = i0 + 2 * (tid % max(ni, 1)), pj = j0 + 2 * (tid / max(ni, 1));
        const int lx = pi - ox, ly = pj - oy;
        q0[c] = on ? ly * PX + COL(lx) : -1;
        qm[c] = ly * PX + COL(lx - 1);
        // right-hand side of the node: straight to a register (issued together with the staging loads below)
        int ri = on ? pi : tx0, rj = on ? pj : ty0;
        if constexpr (WRAP) { ri = image_node(ri, cb.lo[0], cb.hi[0], refl & 1); rj = image_node(rj, cb.lo[1], cb.hi[1], refl & 2); }
        roff[c] = (int)r.off(ri, rj, r.lo[2]);
        rr[c] = r.gp()[roff[c] + (long)(k0 - r.lo[2]) * rks];
        if constexpr (MASK) {
            doff[c] = (int)dmt[fab].off(ri, rj, dmt[fab].lo[2]);
            surf[c] = ri <= cb.lo[0] || ri >= nhi0 || rj <= cb.lo[1] || rj >= nhi1;
            fixed[c] = (surf[c] || k0 <= cb.lo[2] || k0 >= nhi2) ? dmt[fab].gp()[doff[c] + (long)(k0 - dmt[fab].lo[2]) * dks] != 0.0 : false;
        } else { doff[c] = 0; fixed[c] = false; surf[c] = false; }
    }
    // footprint point of this thread in staging round `it`: array offsets (indices clamped into the arrays instead of predicated --
    // footprint points beyond the ghost width are never used by the colour passes) and LDS slot.  32-bit offsets: the launcher
    // checks that the arrays hold fewer than 2^31 values
    constexpr int NLD = (RX * RY + NT - 1) / NT;
    int xoff[NLD], soff[NLD];            // offsets of (xi, xj, plane 0) in the x arrays (xc, xn and xo share one shape) and in sigma
    int lslot[NLD];
    const int xks = x.n[0] * x.n[1], sks = s.n[0] * s.n[1];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int idx = min(tid + it * NT, RX * RY - 1);
        const int lx = idx % RX, ly = idx / RX;
        const int gi = ox + lx, gj = oy + ly;
        int xi, xj, si, sj;
        if constexpr (WRAP) {
            // the box spans the periodic domain: a ghost index is the periodic image of a valid index of the SAME box, so the
            // staging reads the valid data directly and no ghost fill is needed (node hi+1 duplicates node lo)
            xi = image_node(gi, cb.lo[0], cb.hi[0], refl & 1); xj = image_node(gj, cb.lo[1], cb.hi[1], refl & 2);
            si = image_cell(gi, cb.lo[0], cb.hi[0], refl & 1); sj = image_cell(gj, cb.lo[1], cb.hi[1], refl & 2);
        } else {
            xi = min(max(gi, x.lo[0]), x.lo[0] + x.n[0] - 1); xj = min(max(gj, x.lo[1]), x.lo[1] + x.n[1] - 1);
            si = min(max(gi, s.lo[0]), s.lo[0] + s.n[0] - 1); sj = min(max(gj, s.lo[1]), s.lo[1] + s.n[1] - 1);
        }
        xoff[it] = (int)x.off(xi, xj, x.lo[2]);
        soff[it] = CSIG ? 0 : (int)s.off(si, sj, s.lo[2]);
        lslot[it] = (tid + it * NT < RX * RY) ? ly * PX + COL(lx) : -1;
    }
    auto xk = [&](int kk) -> long { if constexpr (WRAP) kk = image_node(kk, cb.lo[2], cb.hi[2], refl & 4); return (long)(kk - x.lo[2]) * xks; };
    auto sk = [&](int kk) -> long { if constexpr (WRAP) kk = image_cell(kk, cb.lo[2], cb.hi[2], refl & 4); return (long)(kk - s.lo[2]) * sks; };
    double* Xm = Xf;                   // plane k-1
    double* Xc = Xf + PL;              // plane k (updated in place)
    double* Xp = Xf + 2 * PL;          // plane k+1
    {
        // stage the first plane: all global loads are issued before the first LDS store
        double v[NLD][CSIG ? 3 : 5];
        const auto *pm = xn.gp() + xk(k0 - 1), *pc = x.gp() + xk(k0), *pp = xn.gp() + xk(k0 + 1);
        const auto *ps0 = s.gp() + sk(k0 - 1), *ps1 = s.gp() + sk(k0);
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            v[it][0] = pm[xoff[it]];
            v[it][1] = pc[xoff[it]];
            v[it][2] = pp[xoff[it]];
            if constexpr (!CSIG) { v[it][3] = ps0[soff[it]]; v[it][4] = ps1[soff[it]]; }
        }
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            if (lslot[it] >= 0) {
                Xm[lslot[it]] = v[it][0]; Xc[lslot[it]] = v[it][1]; Xp[lslot[it]] = v[it][2];
                if constexpr (!CSIG) { Sf[lslot[it]] = v[it][3]; Sf[PL + lslot[it]] = v[it][4]; }
            }
        }
    }
    const int wx = txe - tx0 + 1, wy = tye - ty0 + 1;
    for (int k = k0;; k += 2) {
        const bool has_next = k + 2 <= kend;
        // prefetch of plane k+2 (own plane from xc, its k+3 neighbour from xn, sigma cell planes k+1 and k+2, right-hand side)
        double nv[NLD][CSIG ? 2 : 4], rrn[4];
        bool fixedn[4];
        if (has_next) {
            const auto *pc = x.gp() + xk(k + 2), *pp = xn.gp() + xk(k + 3);
            const auto *ps0 = s.gp() + sk(k + 1), *ps1 = s.gp() + sk(k + 2);
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                nv[it][0] = pc[xoff[it]];
                nv[it][1] = pp[xoff[it]];
                if constexpr (!CSIG) { nv[it][2] = ps0[soff[it]]; nv[it][3] = ps1[soff[it]]; }
            }
            const auto* pr = r.gp() + (long)(k + 2 - r.lo[2]) * rks;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                rrn[c] = pr[roff[c]];
                if constexpr (MASK)
                    fixedn[c] = (surf[c] || k + 2 <= cb.lo[2] || k + 2 >= nhi2) ? dmt[fab].gp()[doff[c] + (long)(k + 2 - dmt[fab].lo[2]) * dks] != 0.0 : false;
                else fixedn[c] = false;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (q0[c] >= 0 && !(MASK && fixed[c])) {
                const int a0 = q0[c], am = qm[c], ap = am + 1;          // row r0: columns c0, cm, cp
                // CSIG: the same expression tree on eight copies of the constant -- bit-identical to the variable-sigma path
                double smmm, spmm, smpm, sppm, smmp, spmp, smpp, sppp;
                if constexpr (CSIG) { smmm = spmm = smpm = sppm = smmp = spmp = smpp = sppp = csig; }
                else {
                    smmm = Sf[am - PX]; spmm = Sf[a0 - PX]; smpm = Sf[am]; sppm = Sf[a0];
                    smmp = Sf[PL + am - PX]; spmp = Sf[PL + a0 - PX]; smpp = Sf[PL + am]; sppp = Sf[PL + a0];
                }
                const double s0 = w.c * (smmm + spmm + smpm + sppm + smmp + spmp + smpp + sppp);
                const double xc = Xc[a0];
                double y = xc * s0;
                y += w.corner * (Xm[am - PX] * smmm + Xm[ap - PX] * spmm + Xm[am + PX] * smpm + Xm[ap + PX] * sppm
                               + Xp[am - PX] * smmp + Xp[ap - PX] * spmp + Xp[am + PX] * smpp + Xp[ap + PX] * sppp);
                y += w.ex * (Xm[a0 - PX] * (smmm + spmm) + Xm[a0 + PX] * (smpm + sppm) + Xp[a0 - PX] * (smmp + spmp) + Xp[a0 + PX] * (smpp + sppp));
                y += w.ey * (Xm[am] * (smmm + smpm) + Xm[ap] * (spmm + sppm) + Xp[am] * (smmp + smpp) + Xp[ap] * (spmp + sppp));
                y += w.ez * (Xc[am - PX] * (smmm + smmp) + Xc[ap - PX] * (spmm + spmp) + Xc[am + PX] * (smpm + smpp) + Xc[ap + PX] * (sppm + sppp));
                y += w.fx * (Xc[am] * (smmm + smpm + smmp + smpp) + Xc[ap] * (spmm + sppm + spmp + sppp));
                y += w.fy * (Xc[a0 - PX] * (smmm + spmm + smmp + spmp) + Xc[a0 + PX] * (smpm + sppm + smpp + sppp));
                y += w.fz * (Xm[a0] * (smmm + spmm + smpm + sppm) + Xp[a0] * (smmp + spmp + smpp + sppp));
                Xc[a0] = xc + (rr[c] - y) / s0;
            }
            __syncthreads();
        }
        for (int idx = tid; idx < wx * wy; idx += NT) {
            const int i = tx0 + idx % wx, j = ty0 + idx / wx;
            xo(i, j, k) = Xc[(j - oy) * PX + COL(i - ox)];
        }
        if (!has_next) break;
        __syncthreads();                 // the write-out above still reads Xc
        // rotate: k+1 becomes the k-1 plane, the prefetched planes take the two freed buffers
        double* t = Xm; Xm = Xp; Xp = Xc; Xc = t;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            if (lslot[it] >= 0) {
                Xc[lslot[it]] = nv[it][0]; Xp[lslot[it]] = nv[it][1];
                if constexpr (!CSIG) { Sf[lslot[it]] = nv[it][2]; Sf[PL + lslot[it]] = nv[it][3]; }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { rr[c] = rrn[c]; fixed[c] = fixedn[c]; }
    }
#undef COL
}

// ------------------------------------------------------------------------------------------------------
// Register-resident form of the plane-fused pass (k_nodal_gsr): the same four colour updates of one k-parity on the same
// shrinking regions as k_nodal_gs4, hence the same doubles, but the planes live in REGISTERS, not in LDS.
//   * A workgroup owns a 64 x 64 footprint of nodes (tile of <= 56 x 56 + the 4-node halo of the recompute) and marches through ppc
//     planes of its parity.  A thread owns a 2 (x) by PB (y) patch of that footprint for the whole march: x at the planes k-1, k, k+1,
//     sigma at the cell planes k-1, k and the right-hand side of its nodes are registers.  The patch holds PB/2 nodes of every in-plane
//     colour, so every lane updates PB/2 nodes in every colour pass (k_nodal_gs4: at most one node per thread and pass, 50-80 % of the lanes).
//   * x-neighbours outside the patch are the adjacent lane's registers: one DPP wave shift (v_mov_b32_dpp wave_shr:1 / wave_shl:1) per
//     dword -- a vector move, no LDS traffic.  A wavefront is two rows of 32 lanes; lanes 0 / 32 (31 / 63) receive a value of the other
//     row, but they own the footprint's first (last) node column, which is read-only.
//   * y-neighbours outside the patch (the row above its first row, needed by the colours with even j, and the row below its last row,
//     needed by the colours with odd j) are the only LDS traffic: every thread publishes the first and last row of its patch -- the
//     static planes k-1, k+1 and sigma once per plane, plane k's first row again after the two even-j passes -- into parity-split rows
//     (conflict-free for the stride-2 access of a colour pass).  9-13 LDS reads per two node updates instead of 35 per node, two barriers
//     per plane instead of five.
//   * the loads of the next plane (two x planes, two sigma planes) are issued before the colour passes of the current one.
// HBM sees each plane once per 56 x 56 tile: footprint / tile = 1.31 (k_nodal_gs4: 40 x 24 / 32 x 16 = 1.875).
__device__ __forceinline__ double gsr_lane_lo(double v)      // the value lane - 1 holds
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);     // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double gsr_lane_hi(double v)      // the value lane + 1 holds
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);     // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// colour-pass updates of a thread that the instruction scheduler may interleave (0: no limit): the update's result passes through an empty
// asm statement every IAMRX_GSR_FENCE nodes, which ends the scheduling region
#ifndef IAMRX_GSR_FENCE
#define IAMRX_GSR_FENCE 2
#endif
#ifndef IAMRX_GSR_EXP
#define IAMRX_GSR_EXP 0     // timing experiments on k_nodal_gsr (wrong results): 1: a product instead of the division, 2: no barriers, 4: no stores,
#endif                      // 8: sigma taken as the constant (no ring reads), 16: no lane shifts, 32: no row exchange through LDS, 64: no sigma loads / ring stores
struct GsrGeom {
    int ntx, nty;       // tiles per box and direction
    int tix, tiy;       // tile pitch (even, <= 56)
    int ppc;            // planes of one parity per workgroup
    int xcd_chunk;      // > 0: XCD-aware order, chunks per tile
    int zc, zn;         // the array the own-parity planes (zc) / the other-parity planes (zn) come from is identically zero: it is not read
    int refl;           // WRAP: bit d: direction d ends on Neumann walls -- mirror images instead of periodic ones (image_node / image_cell)
    int sel;            // 1: only the tiles whose footprint and planes lie inside the box (they read no ghost node of x), 2: only the others, 0: all
};

// the update of one node, k_nodal_gs4's expression tree: x?[db + 1][da + 1] = x(i + da, j + db, plane), s?[db + 1][da + 1] = sigma of
// the cell (i + da, j + db, plane), da, db in {-1, 0}
__device__ __forceinline__ double gsr_update(const NodeW& w, const double (&xm)[3][3], const double (&xc)[3][3], const double (&xp)[3][3],
                                             const double (&sm)[2][2], const double (&sp)[2][2], double rr)
{
    const double smmm = sm[0][0], spmm = sm[0][1], smpm = sm[1][0], sppm = sm[1][1];
    const double smmp = sp[0][0], spmp = sp[0][1], smpp = sp[1][0], sppp = sp[1][1];
    const double s0 = w.c * (smmm + spmm + smpm + sppm + smmp + spmp + smpp + sppp);
    const double x0 = xc[1][1];
    double y = x0 * s0;
    y += w.corner * (xm[0][0] * smmm + xm[0][2] * spmm + xm[2][0] * smpm + xm[2][2] * sppm
                   + xp[0][0] * smmp + xp[0][2] * spmp + xp[2][0] * smpp + xp[2][2] * sppp);
    y += w.ex * (xm[0][1] * (smmm + spmm) + xm[2][1] * (smpm + sppm) + xp[0][1] * (smmp + spmp) + xp[2][1] * (smpp + sppp));
    y += w.ey * (xm[1][0] * (smmm + smpm) + xm[1][2] * (spmm + sppm) + xp[1][0] * (smmp + smpp) + xp[1][2] * (spmp + sppp));
    y += w.ez * (xc[0][0] * (smmm + smmp) + xc[0][2] * (spmm + spmp) + xc[2][0] * (smpm + smpp) + xc[2][2] * (sppm + sppp));
    y += w.fx * (xc[1][0] * (smmm + smpm + smmp + smpp) + xc[1][2] * (spmm + sppm + spmp + sppp));
    y += w.fy * (xc[0][1] * (smmm + spmm + smmp + spmp) + xc[2][1] * (smpm + sppm + smpp + sppp));
    y += w.fz * (xm[1][1] * (smmm + spmm + smpm + sppm) + xp[1][1] * (smmp + spmp + smpp + sppp));
    if (IAMRX_GSR_EXP & 1) return x0 + (rr - y) * s0;
    return x0 + (rr - y) / s0;
}

template <int PB, bool WRAP, bool MASK, bool CSIG>
__global__ void __launch_bounds__(32 * (64 / PB)) k_nodal_gsr(const BoxD* __restrict__ boxes, const FabD* __restrict__ xct, const FabD* __restrict__ xnt,
    const FabD* __restrict__ xot, const FabD* __restrict__ rt, const FabD* __restrict__ st, NodeW w, int kpar, GsrGeom gg,
    const FabD* __restrict__ dmt, double csig)
{
#if defined(__HIP_DEVICE_COMPILE__)    // (the host pass has no global address space: FabD::gp)
    constexpr int NTR = 64 / PB;              // thread rows of the footprint
    constexpr int HX = 34, RP = 2 * HX;       // parity-split row of the 64 columns + one pad column on each side
    constexpr int NSL = NTR + 2;              // row slots: slot q + 1 belongs to thread row q
    constexpr int BUF = NSL * RP;
    // last rows of the patches (read as "row -1" by the thread row below): x planes k-1, k, k+1
    __shared__ double XB[3 * BUF];
    // first rows of the patches (read as "row PB" by the thread row above); written between the two barriers of a plane, read after the second
    __shared__ double XT[3 * BUF];
    // sigma lives in LDS, not in registers: a ring of three cell planes of the footprint (two in use, the third receives plane k + 1 while
    // plane k is smoothed; plane k + 2 replaces k - 1 behind the barrier that ends plane k).  Row r, column c of the footprint sits at
    // (r + 1) * 64 + (c & 1) * 32 + (c >> 1): parity-split columns (conflict-free stride-2 access of a colour pass), one pad row for "row -1"
    // of the first thread row; column -1 of the first lane falls on the last even column of the same row (a value no update uses).
    constexpr int SPL = 65 * 64;
    __shared__ double SG[CSIG ? 1 : 3 * SPL];
    const int fab = blockIdx.y;
    const BoxD cb = boxes[fab];
    int tix, tiy, pk;
    {
        const int nt = gg.ntx * gg.nty;
        int t;
        if (gg.xcd_chunk > 0) {
            // workgroup b runs on XCD b % 8 (speed only): XCD q takes the contiguous range [q, q + 1) * total / 8 of the (chunk, tile)
            // list -- tiles of one z-chunk, which march through the same planes at the same time and share their halo, meet in one L2
            const int q = blockIdx.x & 7, m = blockIdx.x >> 3;
            const int tot = nt * gg.xcd_chunk;
            const int lo = (int)(((long)q * tot) >> 3), cnt = (int)((((long)q + 1) * tot) >> 3) - lo;
            if (m >= cnt) return;
            t = lo + m;
        } else t = blockIdx.x;
        pk = t / nt;
        const int tt = t - pk * nt;
        tix = tt % gg.ntx; tiy = tt / gg.ntx;
    }
    const int kfirst = cb.lo[2] + (((cb.lo[2] & 1) != kpar) ? 1 : 0);
    const int k0 = kfirst + 2 * pk * gg.ppc;
    const int nhi0 = cb.hi[0] + 1, nhi1 = cb.hi[1] + 1, nhi2 = cb.hi[2] + 1;   // last valid node
    // tiles start on even nodes: the in-plane colour of patch node (a, b) is then (a, b & 1) for every thread
    const int tx0 = (cb.lo[0] & ~1) + tix * gg.tix, ty0 = (cb.lo[1] & ~1) + tiy * gg.tiy;
    if (k0 > nhi2 || tx0 > nhi0 || ty0 > nhi1) return;
    const int kend = min(k0 + 2 * (gg.ppc - 1), nhi2);
    const int txs = max(tx0, cb.lo[0]), txe = min(tx0 + gg.tix - 1, nhi0), tys = max(ty0, cb.lo[1]), tye = min(ty0 + gg.tiy - 1, nhi1);
    if (gg.sel != 0) {
        // two-part issue around a ghost exchange (NodalMG::smooth): colour pass c works on the tile grown by 3 - c and reads one node further
        const bool inner = txs - 4 >= cb.lo[0] && txe + 4 <= nhi0 && tys - 4 >= cb.lo[1] && tye + 4 <= nhi1 && k0 - 1 >= cb.lo[2] && kend + 1 <= nhi2;
        if ((gg.sel == 1) != inner) return;
    }
    const FabD x = xct[fab], xn = xnt[fab], xo = xot[fab], r = rt[fab], s = st[fab];
    const int ox = tx0 - 4, oy = ty0 - 4;
    const int tid = threadIdx.x, lane = tid & 63, lx = lane & 31, q = (tid >> 6) * 2 + (lane >> 5);
    // margins: node (a, b) belongs to the region of colour pass c (tile grown by 3 - c) iff min(mx[a], my[b]) >= c; to the tile iff >= 3
    int mx[2], my[PB];
    // byte offsets (unsigned 32 bit: the loads take a uniform plane pointer + a 32-bit lane offset, no 64-bit address registers)
    // MASK: the right-hand side and the Dirichlet mask have the shape of x (the launcher checks): one set of offsets serves the three arrays
    unsigned xcol[2], scol[2], rcol_[MASK ? 1 : 2], xrow[PB], srow[PB], rrow_[MASK ? 1 : PB];
    unsigned (&rcol)[2] = MASK ? xcol : *reinterpret_cast<unsigned (*)[2]>(&rcol_[0]);
    unsigned (&rrow)[PB] = MASK ? xrow : *reinterpret_cast<unsigned (*)[PB]>(&rrow_[0]);
    unsigned (&dcol)[2] = xcol;
    unsigned (&drow)[PB] = xrow;
    unsigned surf = 0;      // MASK: bit b * 2 + a: node on the box surface or outside the box in x or y (only those can be Dirichlet nodes)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int gi = ox + 2 * lx + a;
        mx[a] = min(gi - (txs - 3), (txe + 3) - gi);
        int xi, si;
        if constexpr (WRAP) { xi = image_node(gi, cb.lo[0], cb.hi[0], gg.refl & 1); si = image_cell(gi, cb.lo[0], cb.hi[0], gg.refl & 1); }
        else { xi = min(max(gi, x.lo[0]), x.lo[0] + x.n[0] - 1); si = min(max(gi, s.lo[0]), s.lo[0] + s.n[0] - 1); }
        xcol[a] = 8u * (unsigned)(xi - x.lo[0]);
        scol[a] = CSIG ? 0u : 8u * (unsigned)(si - s.lo[0]);
        if constexpr (!MASK) rcol[a] = 8u * (unsigned)(min(max(xi, r.lo[0]), r.lo[0] + r.n[0] - 1) - r.lo[0]);
        if constexpr (MASK) { if (gi <= cb.lo[0] || gi >= nhi0) surf |= 0x55555555u << a; }
    }
#pragma unroll
    for (int b = 0; b < PB; ++b) {
        const int gj = oy + PB * q + b;
        my[b] = min(gj - (tys - 3), (tye + 3) - gj);
        int xj, sj;
        if constexpr (WRAP) { xj = image_node(gj, cb.lo[1], cb.hi[1], gg.refl & 2); sj = image_cell(gj, cb.lo[1], cb.hi[1], gg.refl & 2); }
        else { xj = min(max(gj, x.lo[1]), x.lo[1] + x.n[1] - 1); sj = min(max(gj, s.lo[1]), s.lo[1] + s.n[1] - 1); }
        xrow[b] = 8u * (unsigned)((xj - x.lo[1]) * x.n[0]);
        srow[b] = CSIG ? 0u : 8u * (unsigned)((sj - s.lo[1]) * s.n[0]);
        if constexpr (!MASK) rrow[b] = 8u * (unsigned)((min(max(xj, r.lo[1]), r.lo[1] + r.n[1] - 1) - r.lo[1]) * r.n[0]);
        if constexpr (MASK) { if (gj <= cb.lo[1] || gj >= nhi1) surf |= 3u << (2 * b); }
    }
    // bit 2 PB c + 2 b + a: node (a, b) belongs to the region of colour pass c (the tile grown by 3 - c nodes); c = 3: to the tile itself
    unsigned um[(8 * PB + 31) / 32];
#pragma unroll
    for (int i = 0; i < (8 * PB + 31) / 32; ++i) um[i] = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int bit = 2 * PB * c + 2 * b + a;
                if (mx[a] >= c && my[b] >= c) um[bit >> 5] |= 1u << (bit & 31);
            }
    const int xks = x.n[0] * x.n[1], sks = s.n[0] * s.n[1], rks = r.n[0] * r.n[1];
    int dks = 0;
    if constexpr (MASK) dks = dmt[fab].n[0] * dmt[fab].n[1];
    auto xk = [&](int kk) -> long { if constexpr (WRAP) kk = image_node(kk, cb.lo[2], cb.hi[2], gg.refl & 4); return (long)(kk - x.lo[2]) * xks; };
    auto sk = [&](int kk) -> long { if constexpr (WRAP) kk = image_cell(kk, cb.lo[2], cb.hi[2], gg.refl & 4); return (long)(kk - s.lo[2]) * sks; };
    // LDS slots of this thread: own row slot q + 1, columns 2 lx (even half) and 2 lx + 1 (odd half); column c sits at
    // ((c + 2) & 1) * HX + ((c + 2) >> 1)
    const int lown = (q + 1) * RP + lx + 1;           // + HX: the odd column
    const int lup = q * RP + lx, ldn = (q + 2) * RP + lx;
    // column 2 lx + t, t in {-1, 0, 1, 2}, of a row whose base (slot * RP + lx) is given
    auto lcol = [](int base, int t) { return base + (t & 1) * HX + 1 + (t >> 1); };

    double Xm[PB][2], Xc[PB][2], Xp[PB][2], R[PB][2];
    unsigned fixedm = 0;                               // MASK: Dirichlet nodes of the current plane
    typedef const __attribute__((address_space(1))) char gbyte;
    auto gat = [](const FabD::gdouble* plane, unsigned byteoff) -> FabD::gdouble& {
        return *(FabD::gdouble*)((gbyte*)plane + (size_t)byteoff);
    };
    auto load_plane = [&](const FabD::gdouble* p, long koff, const unsigned (&col)[2], const unsigned (&row)[PB], double (&dst)[PB][2]) {
        const FabD::gdouble* pp = p + koff;
#pragma unroll
        for (int b = 0; b < PB; ++b) { dst[b][0] = gat(pp, row[b] + col[0]); dst[b][1] = gat(pp, row[b] + col[1]); }
    };
    auto load_mask = [&](int k) -> unsigned {
        unsigned f = 0;
        if constexpr (MASK) {
            const unsigned need = (k <= cb.lo[2] || k >= nhi2) ? 0xffffffffu : surf;
            if (need) {
                const FabD::gdouble* pd = dmt[fab].gp() + (long)(k - dmt[fab].lo[2]) * dks;
#pragma unroll
                for (int b = 0; b < PB; ++b)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        if ((need >> (2 * b + a)) & 1u) f |= (gat(pd, drow[b] + dcol[a]) != 0.0 ? 1u : 0u) << (2 * b + a);
            }
        }
        return f;
    };
    // the mask of the next plane, as loaded: issued with the other loads in the middle of an iteration, turned into bits at its end (a load
    // compared on the spot makes the wavefront wait for it and for every load issued before it)
    double Mr[MASK ? PB : 1][2];
    auto fetch_mask = [&](int k) {
        if constexpr (MASK) {
            const unsigned need = (k <= cb.lo[2] || k >= nhi2) ? 0xffffffffu : surf;
            const FabD::gdouble* pd = dmt[fab].gp() + (long)(k - dmt[fab].lo[2]) * dks;
#pragma unroll
            for (int b = 0; b < PB; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    Mr[b][a] = 0.0;
                    if ((need >> (2 * b + a)) & 1u) Mr[b][a] = gat(pd, drow[b] + dcol[a]);
                }
        }
    };
    auto mask_bits = [&]() -> unsigned {
        unsigned f = 0;
        if constexpr (MASK) {
#pragma unroll
            for (int b = 0; b < PB; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) f |= (Mr[b][a] != 0.0 ? 1u : 0u) << (2 * b + a);
        }
        return f;
    };
    auto load_rhs = [&](int k, int par) {
        const FabD::gdouble* pr = r.gp() + (long)(k - r.lo[2]) * rks;
#pragma unroll
        for (int b = par; b < PB; b += 2) {
            R[b][0] = gat(pr, rrow[b] + rcol[0]); R[b][1] = gat(pr, rrow[b] + rcol[1]);
        }
    };
    // sigma ring: element of the own cell (a, b) of ring slot sl: sl * SPL + sgo + (b + 1) * 64 + a * 32
    const int sgo = PB * q * 64 + lx;
    auto ring_store = [&](int sl, const double (&T)[PB][2]) {
        double* d = SG + sl * SPL + sgo;
#pragma unroll
        for (int b = 0; b < PB; ++b) { d[(b + 1) * 64] = T[b][0]; d[(b + 1) * 64 + 32] = T[b][1]; }
    };
    int slm = 0, slp = 1, slf = 2;                     // ring slots of the sigma planes k - 1, k and the free one
    double T1[PB][2];                                  // a sigma plane on its way from HBM to the ring
    auto zero_plane = [&](double (&dst)[PB][2]) {
#pragma unroll
        for (int b = 0; b < PB; ++b) dst[b][0] = dst[b][1] = 0.0;
    };
    if (gg.zn) { zero_plane(Xm); zero_plane(Xp); } else { load_plane(xn.gp(), xk(k0 - 1), xcol, xrow, Xm); load_plane(xn.gp(), xk(k0 + 1), xcol, xrow, Xp); }
    if (gg.zc) zero_plane(Xc); else load_plane(x.gp(), xk(k0), xcol, xrow, Xc);
    if constexpr (!CSIG) {
        double T0[PB][2];
        load_plane(s.gp(), sk(k0 - 1), scol, srow, T0); load_plane(s.gp(), sk(k0), scol, srow, T1);
        ring_store(slm, T0); ring_store(slp, T1);
    }
    load_rhs(k0, 0); load_rhs(k0, 1);
    fixedm = load_mask(k0);
    double exp_sink = 0.0;      // (IAMRX_GSR_EXP & 4: keeps the arithmetic alive without the stores)
    for (int k = k0;; k += 2) {
        const bool has_next = k + 2 <= kend;
        // prefetch of plane k + 2 in two halves: the sigma cell plane k + 1 now; sigma k + 2 and the x planes k + 2, k + 3 after the even-j passes
        double Nc[PB][2], Np[PB][2];
        if constexpr (!CSIG) { if (has_next) load_plane(s.gp(), sk(k + 1), scol, srow, T1); }
        // publish the patch's last row (planes k-1, k, k+1)
        XB[lown] = Xm[PB - 1][0]; XB[lown + HX] = Xm[PB - 1][1];
        XB[BUF + lown] = Xc[PB - 1][0]; XB[BUF + lown + HX] = Xc[PB - 1][1];
        XB[2 * BUF + lown] = Xp[PB - 1][0]; XB[2 * BUF + lown + HX] = Xp[PB - 1][1];
        if (!(IAMRX_GSR_EXP & 2)) __syncthreads();
        const double* SGm = SG + (CSIG ? 0 : slm * SPL + sgo);
        const double* SGp = SG + (CSIG ? 0 : slp * SPL + sgo);
        auto pass = [&](auto cxc, auto cyc) {
            constexpr int CX = decltype(cxc)::value, CY = decltype(cyc)::value, c = CX + 2 * CY;
            // the column outside the patch on the side of the nodes of this pass is the adjacent lane's: one DPP shift per value, on demand
            auto fx = [&](const double (&P)[PB][2], int bb) { if constexpr (IAMRX_GSR_EXP & 16) return P[bb][1 - CX]; else if constexpr (CX == 0) return gsr_lane_lo(P[bb][1]); else return gsr_lane_hi(P[bb][0]); };
            // the row outside the patch: LDS (row -1 for the even-j colours, row PB for the odd-j ones)
            double em[3], ec[3], ep[3];
            {
                const int base = CY == 0 ? lup : ldn;
                const double* Bx = CY == 0 ? XB : XT;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int o = lcol(base, CX + d - 1);
                    if (IAMRX_GSR_EXP & 32) { em[d] = Xm[0][0]; ec[d] = Xc[0][0]; ep[d] = Xp[0][0]; } else { em[d] = Bx[o]; ec[d] = Bx[BUF + o]; ep[d] = Bx[2 * BUF + o]; }
                }
            }
#pragma unroll
            for (int n_ = 0; n_ < PB / 2; ++n_) {
                // the node next to the row that came from LDS first: its values die with it
                const int n = CY == 0 ? n_ : PB / 2 - 1 - n_;
                const int b = CY + 2 * n;
                double vm[3][3], vc[3][3], vp[3][3], tm[2][2], tp[2][2];
#pragma unroll
                for (int db = -1; db <= 1; ++db)
#pragma unroll
                    for (int da = -1; da <= 1; ++da) {
                        const int bb = b + db, aa = CX + da;
                        double m_, c_, p_;
                        if (bb < 0 || bb >= PB) { m_ = em[da + 1]; c_ = ec[da + 1]; p_ = ep[da + 1]; }
                        else if (aa < 0 || aa > 1) { m_ = fx(Xm, bb); c_ = fx(Xc, bb); p_ = fx(Xp, bb); }
                        else { m_ = Xm[bb][aa]; c_ = Xc[bb][aa]; p_ = Xp[bb][aa]; }
                        vm[db + 1][da + 1] = m_; vc[db + 1][da + 1] = c_; vp[db + 1][da + 1] = p_;
                    }
#pragma unroll
                for (int db = -1; db <= 0; ++db)
#pragma unroll
                    for (int da = -1; da <= 0; ++da) {
                        if constexpr (CSIG || (IAMRX_GSR_EXP & 8)) { tm[db + 1][da + 1] = csig; tp[db + 1][da + 1] = csig; }
                        else {
                            const int aa = CX + da;        // cell column -1, 0 or 1 of the patch
                            const int o = (b + db + 1) * 64 + (aa < 0 ? 31 : aa * 32);
                            tm[db + 1][da + 1] = SGm[o]; tp[db + 1][da + 1] = SGp[o];
                        }
                    }
                double nv = gsr_update(w, vm, vc, vp, tm, tp, R[b][CX]);
                // evaluated by every lane: a branch around the arithmetic would keep all lane-shifted operands (which must be formed
                // under the full EXEC mask) alive across it; the fence also keeps the updates of a pass from being interleaved
                // beyond what the register file holds
#if IAMRX_GSR_FENCE > 0
                if ((n_ % IAMRX_GSR_FENCE) == IAMRX_GSR_FENCE - 1) asm volatile("" : "+v"(nv));
#endif
                const int ubit = 2 * PB * c + 2 * b + CX;
                bool upd = ((um[ubit >> 5] >> (ubit & 31)) & 1u) != 0;
                if constexpr (MASK) upd = upd && !((fixedm >> (2 * b + CX)) & 1u);
                Xc[b][CX] = upd ? nv : Xc[b][CX];
            }
        };
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        // the patch's first row: plane k now holds its colour-0 and colour-1 values, which the odd-j passes of the thread row above read.
        // (every wavefront is past the first barrier: nobody reads the first rows of the previous plane any more)
        XT[lown] = Xm[0][0]; XT[lown + HX] = Xm[0][1];
        XT[BUF + lown] = Xc[0][0]; XT[BUF + lown + HX] = Xc[0][1];
        XT[2 * BUF + lown] = Xp[0][0]; XT[2 * BUF + lown + HX] = Xp[0][1];
        if (has_next) {
            // right-hand side of the next plane: the even rows are free now, the odd rows after the odd-j passes
            load_rhs(k + 2, 0);
            if constexpr (!CSIG) {
                ring_store(slf, T1);                                  // sigma plane k + 1: into the free slot
                load_plane(s.gp(), sk(k + 2), scol, srow, T1);
            }
            if (gg.zc) zero_plane(Nc); else load_plane(x.gp(), xk(k + 2), xcol, xrow, Nc);
            if (gg.zn) zero_plane(Np); else load_plane(xn.gp(), xk(k + 3), xcol, xrow, Np);
            fetch_mask(k + 2);
        }
        if (!(IAMRX_GSR_EXP & 2)) __syncthreads();
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        if (has_next) load_rhs(k + 2, 1);
        {
            // (16-byte stores for the threads whose whole patch lies in the tile, without a branch per node, were measured: 107.3 against
            // 108.7 us and 256 VGPRs + scratch in the ghost-filled variant -- the stores cost what their 68 MB cost, not their form)
            FabD::gdouble* po = xo.gp() + xk(k);
#pragma unroll
            for (int b = 0; b < PB; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if ((um[(2 * PB * 3 + 2 * b + a) >> 5] >> ((2 * PB * 3 + 2 * b + a) & 31)) & 1u) {
                        if (!(IAMRX_GSR_EXP & 4)) gat(po, xrow[b] + xcol[a]) = Xc[b][a]; else exp_sink += Xc[b][a];
                    }
        }
        if (!has_next) break;
        // every wavefront is done with the sigma planes k - 1, k and the first rows of this plane
        if (!(IAMRX_GSR_EXP & 2)) __syncthreads();
        if constexpr (!CSIG) ring_store(slm, T1);                     // sigma plane k + 2 replaces k - 1
        { const int t = slm; slm = slf; slf = slp; slp = t; }
        // rotate: k + 1 becomes the k - 1 plane
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a) { Xm[b][a] = Xp[b][a]; Xc[b][a] = Nc[b][a]; Xp[b][a] = Np[b][a]; }
        fixedm = mask_bits();
    }
    if ((IAMRX_GSR_EXP & 4) && exp_sink == 1.2345e300) gat(xo.gp(), 0u) = exp_sink;
#endif
}

// HIP-event probes around selected kernel launches (kernels.h: kernel_probe_*), on the launch stream, so that the roofline figures of the
// dominant kernels are measured inside the running time step (bench.py)
namespace {
struct KProbe {
    bool on = false;
    long min_points = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    long seen = 0;
    int stride = 1;          // every stride-th qualifying launch is timed (keeps the event overhead out of the step time)
};
KProbe g_probe[PROBE_COUNT];
}
void kernel_probe_start(int which, long min_points, int stride)
{
    KProbe& pb = g_probe[which];
    pb.on = true; pb.min_points = min_points; pb.used = 0; pb.seen = 0; pb.stride = stride < 1 ? 1 : stride;
}
void kernel_probe_stop(int which, double* total_ms, long* launches)
{
    KProbe& pb = g_probe[which];
    Context::get().sync();
    double ms = 0.0;
    for (size_t i = 0; i < pb.used; ++i) {
        float t = 0.f;
        IAMRX_HIP_CHECK(hipEventElapsedTime(&t, pb.ev[i].first, pb.ev[i].second));
        ms += t;
    }
    *total_ms = ms; *launches = (long)pb.used;
    pb.on = false; pb.used = 0;
}
static bool g_probes_paused = false;
void kernel_probes_pause(bool on) { g_probes_paused = on; }
bool kernel_probe_begin(int which, long points)
{
    KProbe& pb = g_probe[which];
    if (g_probes_paused) return false;
    if (!(pb.on && points >= pb.min_points && (pb.seen++ % pb.stride) == 0)) return false;
    if (pb.used == pb.ev.size()) {
        hipEvent_t e0, e1;
        IAMRX_HIP_CHECK(hipEventCreate(&e0)); IAMRX_HIP_CHECK(hipEventCreate(&e1));
        pb.ev.emplace_back(e0, e1);
    }
    IAMRX_HIP_CHECK(hipEventRecord(pb.ev[pb.used].first, Context::get().stream));
    return true;
}
void kernel_probe_end(int which)
{
    KProbe& pb = g_probe[which];
    IAMRX_HIP_CHECK(hipEventRecord(pb.ev[pb.used].second, Context::get().stream));
    ++pb.used;
}
void gs4_probe_start(long min_nodes, int stride) { kernel_probe_start(PROBE_NODAL_GS4, min_nodes, stride); }
void gs4_probe_stop(double* total_ms, long* launches) { kernel_probe_stop(PROBE_NODAL_GS4, total_ms, launches); }

template <int TX, int TY, int NT>
static void gs4_launch(const Layout& l, const Geometry& g, const MultiFab& xc, const MultiFab& xn, MultiFab& xo, const MultiFab& rhs, const MultiFab& sig,
                       int kpar, bool wrap, const MultiFab* dmask, const double* csig, int refl)
{
    // (a "fat" last tile that takes the single leftover node column of a box of n = k TX cells along -- 8 x 16 instead of 9 x 17 tiles at 257^3
    // nodes -- was measured: the two extra footprint columns / rows cost more (253 us) than the empty tiles (223 us))
    const int ntx = (l.max_len[0] + 1 + TX - 1) / TX, nty = (l.max_len[1] + 1 + TY - 1) / TY, npl_all = (l.max_len[2] + 1 + 1) / 2 + 1;
    const int nt = ntx * nty;
    // planes per workgroup (z-march): as long as the launch still fills the chip about twice over (256 CUs x 4-6 resident workgroups)
    const int ppc_env = (int)tune("GS4_PPC", 0);
    const long wg_target = (long)tune("GS4_WGS", 2048);
    int ppc = ppc_env > 0 ? ppc_env : (int)std::max<long>(1, std::min<long>(16, (long)nt * l.nlocal() * npl_all / wg_target));
    const int npl = (npl_all + ppc - 1) / ppc;             // chunks per tile
    for (int f = 0; f < l.nlocal(); ++f) {                  // the kernel indexes with 32-bit offsets
        const BoxD& b = l.boxes[l.local[f]];
        IAMRX_ASSERT((long)(b.len(0) + 1 + 2 * xc.ngrow) * (b.len(1) + 1 + 2 * xc.ngrow) * (b.len(2) + 1 + 2 * xc.ngrow) < 2147483647L);
    }
    const bool xcd_aware = tune("XCD_AWARE", 1) != 0;
    int xcd_chunk = 0;
    unsigned gx = (unsigned)(nt * npl);
    if (xcd_aware && nt >= 16) {
        xcd_chunk = npl;                                   // planes per tile
        const int maxcnt = (nt + 7) / 8;                   // tiles of the largest slab
        gx = 8u * (unsigned)(maxcnt * npl);
    }
    dim3 grid(gx, (unsigned)l.nlocal());
    const bool rec = kernel_probe_begin(PROBE_NODAL_GS4, (long)(l.max_len[0] + 1) * (l.max_len[1] + 1) * (l.max_len[2] + 1));
#define IAMRX_GS4(W, M, C) hipLaunchKernelGGL((k_nodal_gs4<TX, TY, NT, W, M, C>), grid, dim3(NT), 0, Context::get().stream, l.d_boxes, xc.d_tab, xn.d_tab, \
                                             xo.d_tab, rhs.d_tab, sig.d_tab, make_w(g), kpar, ntx, nty, xcd_chunk, dmask ? dmask->d_tab : nullptr, csig ? *csig : 0.0, ppc, refl)
    if (dmask) {
        IAMRX_ASSERT(!wrap && dmask->ngrow >= 3 && !csig);
        IAMRX_GS4(false, true, false);
    } else if (wrap) { if (csig) IAMRX_GS4(true, false, true); else IAMRX_GS4(true, false, false); }
    else { if (csig) IAMRX_GS4(false, false, true); else IAMRX_GS4(false, false, false); }
#undef IAMRX_GS4
    if (rec) kernel_probe_end(PROBE_NODAL_GS4);
}

// tiles of k_nodal_gsr: the fewest tiles of at most 56 nodes that cover the largest box (starting on an even node), equal even pitch
static void gsr_tiles(int len_nodes, int& nt, int& pitch)
{
    const int span = len_nodes + 1;               // the first tile starts on the even node <= lo
    nt = (span + 55) / 56;
    pitch = (span + nt - 1) / nt;
    pitch += pitch & 1;
}

template <int PB>
static void gsr_launch(const Layout& l, const Geometry& g, const MultiFab& xc, const MultiFab& xn, MultiFab& xo, const MultiFab& rhs, const MultiFab& sig,
                       int kpar, bool wrap, const MultiFab* dmask, const double* csig, int zero_flags, int refl, int sel, hipStream_t on)
{
    GsrGeom gg;
    gg.sel = sel;
    hipStream_t strm = on ? on : Context::get().stream;
    gg.zc = zero_flags & 1; gg.zn = (zero_flags >> 1) & 1;
    gg.refl = wrap ? refl : 0;
    gsr_tiles(l.max_len[0] + 1, gg.ntx, gg.tix);
    gsr_tiles(l.max_len[1] + 1, gg.nty, gg.tiy);
    const int npl_all = (l.max_len[2] + 1 + 1) / 2 + 1;
    const int nt = gg.ntx * gg.nty;
    // z-chunks: one resident workgroup per CU (PB = 4: 8 wavefronts of <= 256 registers) and ONE round of workgroups -- a second, partly
    // filled round would cost as much as the first
    const long slots = tune("GSR_SLOTS", 256);
    int npl = (int)tune("GSR_NPL", 0);
    if (npl <= 0) npl = (int)std::max<long>(1, std::min<long>(npl_all, slots / ((long)nt * l.nlocal())));
    gg.ppc = (npl_all + npl - 1) / npl;
    npl = (npl_all + gg.ppc - 1) / gg.ppc;
    for (int f = 0; f < l.nlocal(); ++f) {                  // the kernel indexes with 32-bit offsets
        const BoxD& b = l.boxes[l.local[f]];
        IAMRX_ASSERT((long)(b.len(0) + 1 + 2 * xc.ngrow) * (b.len(1) + 1 + 2 * xc.ngrow) * (b.len(2) + 1 + 2 * xc.ngrow) < 2147483647L);
    }
    const long total = (long)nt * npl;
    unsigned gx = (unsigned)total;
    gg.xcd_chunk = 0;
    if (tune("XCD_AWARE", 1) != 0 && total >= 16) { gg.xcd_chunk = npl; gx = 8u * (unsigned)((total + 7) / 8); }
    dim3 grid(gx, (unsigned)l.nlocal());
    constexpr int NT = 32 * (64 / PB);
    const bool rec = sel == 0 && !on && kernel_probe_begin(PROBE_NODAL_GS4, (long)(l.max_len[0] + 1) * (l.max_len[1] + 1) * (l.max_len[2] + 1));
#define IAMRX_GSR(W, M, C) hipLaunchKernelGGL((k_nodal_gsr<PB, W, M, C>), grid, dim3(NT), 0, strm, l.d_boxes, xc.d_tab, xn.d_tab, \
                                             xo.d_tab, rhs.d_tab, sig.d_tab, make_w(g), kpar, gg, dmask ? dmask->d_tab : nullptr, csig ? *csig : 0.0)
    if (dmask) {
        IAMRX_ASSERT(!wrap && dmask->ngrow >= 3 && !csig);
        IAMRX_GSR(false, true, false);                                  // (same shape of x, rhs and mask: checked by nodal_gs_fused_pass)
    } else if (wrap) { if (csig) IAMRX_GSR(true, false, true); else IAMRX_GSR(true, false, false); }
    else { if (csig) IAMRX_GSR(false, false, true); else IAMRX_GSR(false, false, false); }
#undef IAMRX_GSR
    if (rec) kernel_probe_end(PROBE_NODAL_GS4);
}

// one k-parity pass (kpar = 0: colours 0-3, kpar = 1: colours 4-7); wrap: see periodic_wrap_ok; needs x.ngrow >= 4, sig.ngrow >= 4, rhs.ngrow >= 3
// true if the level is one box that spans a fully periodic domain: the smoother kernels can then read periodic images straight from
// the valid data (wrap = true) and the ghost fills of x / rhs in front of it can be skipped
bool periodic_wrap_ok(const Geometry& g, const Layout& l, int min_len)
{
    const bool on = tune("PERIODIC_WRAP", 1) != 0;
    if (!on || l.boxes.size() != 1 || l.nlocal() != 1) return false;
    for (int d = 0; d < 3; ++d)
        if (!g.periodic[d] || l.boxes[0].lo[d] != g.domain.lo[d] || l.boxes[0].hi[d] != g.domain.hi[d] || l.boxes[0].len(d) < min_len) return false;
    return true;
}

// ... or one box that spans a domain whose non-periodic directions end on Neumann walls on both sides (the pressure of a closed or
// channel-like domain: LidDrivenCavity): the same kernels with mirror images in those directions (refl: bit d).  No Dirichlet mask (the
// caller checks).  IAMRX_NODAL_REFLECT_WRAP (1): 0 = ghost fills (nodal_reflect_bc) in front of every pass.
bool nodal_wrap_or_reflect_ok(const Geometry& g, const Layout& l, const DomainBC& bc, int min_len, int* refl)
{
    *refl = 0;
    if (periodic_wrap_ok(g, l, min_len)) return true;
    if (tune("PERIODIC_WRAP", 1) == 0 || tune("NODAL_REFLECT_WRAP", 1) == 0 || l.boxes.size() != 1 || l.nlocal() != 1) return false;
    int r = 0;
    for (int d = 0; d < 3; ++d) {
        if (l.boxes[0].lo[d] != g.domain.lo[d] || l.boxes[0].hi[d] != g.domain.hi[d] || l.boxes[0].len(d) < min_len) return false;
        if (g.periodic[d]) continue;
        if (bc.lo[d] != lo_neumann || bc.hi[d] != lo_neumann) return false;
        r |= 1 << d;
    }
    *refl = r;
    return true;
}

// the register-resident kernel takes this level (and honours zero_flags: bit 0 / 1 = xc / xn is identically zero and need not be read)
bool nodal_gsr_applies(const MultiFab& x, const MultiFab& rhs, const MultiFab* dmask)
{
    const Layout& l = *x.layout;
    const bool mask_shapes_ok = !dmask || (dmask->ngrow == x.ngrow && rhs.ngrow == x.ngrow);
    return tune("GSR", 1) != 0 && l.max_len[0] >= tune("GSR_MIN", 48) && l.max_len[1] >= tune("GSR_MIN", 48) && mask_shapes_ok;
}

// the level is smoothed by k_nodal_gsr and its boxes hold tiles whose footprint lies inside the box (>= 3 tiles in x and y: 113 nodes)
bool nodal_gsr_splits(const MultiFab& x, const MultiFab& rhs, const MultiFab* dmask)
{
    if (x.nlocal() == 0 || !nodal_gsr_applies(x, rhs, dmask)) return false;
    const Layout& l = *x.layout;
    int ntx, nty, p;
    gsr_tiles(l.max_len[0] + 1, ntx, p);
    gsr_tiles(l.max_len[1] + 1, nty, p);
    return ntx >= 3 && nty >= 3 && l.max_len[2] >= 16;
}

void nodal_gs_fused_pass(const Geometry& g, const MultiFab& xc, const MultiFab& xn, MultiFab& xo, const MultiFab& rhs, const MultiFab& sig, int kpar, bool wrap,
                         const MultiFab* dmask, const double* csig, int zero_flags, int refl, int sel, hipStream_t on)
{
    IAMRX_ASSERT(refl == 0 || wrap);
    const MultiFab& x = xc;
    if (x.nlocal() == 0) return;
    IAMRX_ASSERT(x.ngrow >= 4 && sig.ngrow >= 4 && rhs.ngrow >= 3 && xn.ngrow == x.ngrow && xo.ngrow == x.ngrow && xo.d_tab != xc.d_tab);
    const Layout& l = *x.layout;
    // tile shape: 32x16 nodes / 256 threads (40x24 footprint, 40 KB of LDS: 4 workgroups = 16 waves per CU).  Measured at 256^3
    // on MI355X: 0.151 ms per launch against 0.179 ms for 32x32 / 512 threads (IAMRX_GS4_TILE=1; fewer redundant loads and
    // updates but 8-wave barriers)
    // levels whose boxes are at least GSR_MIN cells long in x and y: the register-resident kernel (IAMRX_GSR=0: k_nodal_gs4 everywhere)
    if (nodal_gsr_applies(x, rhs, dmask)) {
        if (tune("GSR_PB", 4) == 8) gsr_launch<8>(l, g, xc, xn, xo, rhs, sig, kpar, wrap, dmask, csig, zero_flags, refl, sel, on);
        else gsr_launch<4>(l, g, xc, xn, xo, rhs, sig, kpar, wrap, dmask, csig, zero_flags, refl, sel, on);
        return;
    }
    IAMRX_ASSERT(zero_flags == 0 && sel == 0 && !on);       // k_nodal_gs4 reads its inputs; the two-part issue is k_nodal_gsr's
    const int big = (int)tune("GS4_TILE", 0);
    if (big && l.max_len[1] + 1 > 16) gs4_launch<32, 32, 512>(l, g, xc, xn, xo, rhs, sig, kpar, wrap, dmask, csig, wrap ? refl : 0);
    else gs4_launch<32, 16, 256>(l, g, xc, xn, xo, rhs, sig, kpar, wrap, dmask, csig, wrap ? refl : 0);
}

// ------------------------------------------------------------------------------------------------------
// Coarse-level smoother: ALL sweeps x 8 colours of a small single-box, fully periodic level in ONE launch of
// one workgroup (periodic images are taken by index wrap, colours are separated by __syncthreads).  Replaces
// nsweeps*8*(ghost fill + colour kernel) launches whose cost on <= 32^3 levels is pure launch latency.
// Same arithmetic and ordering as the general path (the wrapped neighbour IS the ghost value).
template <class XA>
__device__ __forceinline__ double node_Ax_wrap(const XA& x, const FabD& s, const NodeW& w, int im, int i, int ip, int jm, int j, int jp,
                                               int km, int k, int kp, double& s0)
{
    const double smmm = s(im, jm, km), spmm = s(i, jm, km), smpm = s(im, j, km), sppm = s(i, j, km);
    const double smmp = s(im, jm, k), spmp = s(i, jm, k), smpp = s(im, j, k), sppp = s(i, j, k);
    s0 = w.c * (smmm + spmm + smpm + sppm + smmp + spmp + smpp + sppp);
    double y = x(i, j, k) * s0;
    y += w.corner * (x(im, jm, km) * smmm + x(ip, jm, km) * spmm + x(im, jp, km) * smpm + x(ip, jp, km) * sppm
                   + x(im, jm, kp) * smmp + x(ip, jm, kp) * spmp + x(im, jp, kp) * smpp + x(ip, jp, kp) * sppp);
    y += w.ex * (x(i, jm, km) * (smmm + spmm) + x(i, jp, km) * (smpm + sppm) + x(i, jm, kp) * (smmp + spmp) + x(i, jp, kp) * (smpp + sppp));
    y += w.ey * (x(im, j, km) * (smmm + smpm) + x(ip, j, km) * (spmm + sppm) + x(im, j, kp) * (smmp + smpp) + x(ip, j, kp) * (spmp + sppp));
    y += w.ez * (x(im, jm, k) * (smmm + smmp) + x(ip, jm, k) * (spmm + spmp) + x(im, jp, k) * (smpm + smpp) + x(ip, jp, k) * (sppm + sppp));
    y += w.fx * (x(im, j, k) * (smmm + smpm + smmp + smpp) + x(ip, j, k) * (spmm + sppm + spmp + sppp));
    y += w.fy * (x(i, jm, k) * (smmm + spmm + smmp + spmp) + x(i, jp, k) * (smpm + sppm + smpp + sppp));
    y += w.fz * (x(i, j, km) * (smmm + spmm + smpm + sppm) + x(i, j, kp) * (smmp + spmp + smpp + sppp));
    return y;
}

__global__ void __launch_bounds__(1024) k_nodal_smooth_small(const FabD* __restrict__ xt, const FabD* __restrict__ rt,
    const FabD* __restrict__ st, NodeW w, int n0, int n1, int n2, int lo0, int lo1, int lo2, int nsweeps)
{
    const FabD x = xt[0], r = rt[0], s = st[0];
    const int h0 = n0 >> 1, h1 = n1 >> 1, h2 = n2 >> 1;
    const int nlat = h0 * h1 * h2;
    for (int ns = 0; ns < nsweeps; ++ns) {
        for (int c = 0; c < 8; ++c) {
            const int cx = c & 1, cy = (c >> 1) & 1, cz = (c >> 2) & 1;
            for (int idx = threadIdx.x; idx < nlat; idx += 1024) {
                const int a = idx % h0, q = idx / h0;
                const int b = q % h1, d = q / h1;
                // unique nodes 0..n-1 relative to the box origin; absolute parity = (lo + rel) & 1, lo is even for a coarsenable box
                const int ri = 2 * a + ((cx - lo0) & 1), rj = 2 * b + ((cy - lo1) & 1), rk = 2 * d + ((cz - lo2) & 1);
                const int i = lo0 + ri, j = lo1 + rj, k = lo2 + rk;
                const int im = lo0 + (ri == 0 ? n0 - 1 : ri - 1), ip = lo0 + (ri == n0 - 1 ? 0 : ri + 1);
                const int jm = lo1 + (rj == 0 ? n1 - 1 : rj - 1), jp = lo1 + (rj == n1 - 1 ? 0 : rj + 1);
                const int km = lo2 + (rk == 0 ? n2 - 1 : rk - 1), kp = lo2 + (rk == n2 - 1 ? 0 : rk + 1);
                double s0;
                const double Ax = node_Ax_wrap(x, s, w, im, i, ip, jm, j, jp, km, k, kp, s0);
                x(i, j, k) += (r(i, j, k) - Ax) / s0;
            }
            __syncthreads();
        }
    }
    // periodic duplicates (index n) take the owner's value; ghosts are filled by the caller's FillBoundary
    const int m0 = n0 + 1, m1 = n1 + 1, m2 = n2 + 1;
    for (int idx = threadIdx.x; idx < m0 * m1 * m2; idx += 1024) {
        const int ri = idx % m0, q = idx / m0;
        const int rj = q % m1, rk = q / m1;
        if (ri == n0 || rj == n1 || rk == n2)
            x(lo0 + ri, lo1 + rj, lo2 + rk) = x(lo0 + (ri == n0 ? 0 : ri), lo1 + (rj == n1 ? 0 : rj), lo2 + (rk == n2 ? 0 : rk));
    }
}

// returns false if the level does not qualify (then the caller uses the general path)
bool nodal_smooth_small(const Geometry& g, MultiFab& x, const MultiFab& rhs, const MultiFab& sig, int nsweeps)
{
    const Layout& l = *x.layout;
    if (l.boxes.size() != 1 || l.nlocal() != 1) return false;
    const BoxD& b = l.boxes[0];
    long cells = 1;
    for (int d = 0; d < 3; ++d) {
        if (!g.periodic[d] || b.lo[d] != g.domain.lo[d] || b.hi[d] != g.domain.hi[d] || (b.len(d) & 1)) return false;
        cells *= b.len(d);
    }
    if (cells > 8L * 8 * 8) return false;     // measured: one workgroup wins up to 8^3 (13 us/sweep), loses from 32^3 on
    hipLaunchKernelGGL(k_nodal_smooth_small, dim3(1), dim3(1024), 0, Context::get().stream, x.d_tab, rhs.d_tab, sig.d_tab, make_w(g),
                       b.len(0), b.len(1), b.len(2), b.lo[0], b.lo[1], b.lo[2], nsweeps);
    return true;
}

// ------------------------------------------------------------------------------------------------------
// Bottom solve of a small single-box, fully periodic level on the device (the nodal counterpart of k_abec_bottom, k_abec.hip):
// NodalMG::vcycle's bottom block -- BiCGStab to bottom_reltol as NodalMG::bicgstab drives it from the host, then nub (or, after a
// breakdown, nuf + nuf) smooth calls of nsweeps 8-colour Gauss-Seidel sweeps -- in ONE launch of ONE workgroup: one unique node per
// thread, vectors in registers, the vector the operator is applied to in LDS with index wrap (node_Ax_wrap: the wrapped neighbour
// IS the periodic ghost value), reductions inside the workgroup.  No host synchronisation, no further launches.
constexpr int NBOT_NT = 512;
struct LdsNodes {
    const double* p; int n0, n1, lo0, lo1, lo2;
    __device__ __forceinline__ double operator()(int i, int j, int k) const { return p[(i - lo0) + n0 * ((j - lo1) + n1 * (k - lo2))]; }
};
__device__ __forceinline__ double nb_sum(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < NBOT_NT / 64; ++w) s += red[w];
    return s;
}
__device__ __forceinline__ double nb_max(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < NBOT_NT / 64; ++w) s = fmax(s, red[w]);
    return s;
}

__global__ void __launch_bounds__(NBOT_NT) k_nodal_bottom(const FabD* __restrict__ cort, const FabD* __restrict__ rest, const FabD* __restrict__ sigt,
    NodeW w, int n0, int n1, int n2, int lo0, int lo1, int lo2, int singular, double eps_rel, int maxiter, int nsweeps, int nub, int nuf,
    int* __restrict__ iters_out)
{
    __shared__ double V[NBOT_NT];
    __shared__ double red[NBOT_NT / 64];
    const FabD cor = cort[0], res = rest[0], s = sigt[0];
    const int N = n0 * n1 * n2, tid = threadIdx.x;
    const bool on = tid < N;
    const int ri = on ? tid % n0 : 0, rj = on ? (tid / n0) % n1 : 0, rk = on ? tid / (n0 * n1) : 0;
    const int i = lo0 + ri, j = lo1 + rj, k = lo2 + rk;
    const int im = lo0 + (ri == 0 ? n0 - 1 : ri - 1), ip = lo0 + (ri == n0 - 1 ? 0 : ri + 1);
    const int jm = lo1 + (rj == 0 ? n1 - 1 : rj - 1), jp = lo1 + (rj == n1 - 1 ? 0 : rj + 1);
    const int km = lo2 + (rk == 0 ? n2 - 1 : rk - 1), kp = lo2 + (rk == n2 - 1 ? 0 : rk + 1);
    const LdsNodes A{V, n0, n1, lo0, lo1, lo2};
    const double rhs0 = on ? (double)res(i, j, k) : 0.0;
    double s0 = 1.0;
    auto apply = [&](double xv) -> double {
        __syncthreads();
        if (on) V[tid] = xv;
        __syncthreads();
        if (!on) return 0.0;
        return node_Ax_wrap(A, s, w, im, i, ip, jm, j, jp, km, k, kp, s0);
    };
    double bb = rhs0;
    if (singular) bb -= nb_sum(rhs0, red) / (double)N;
    if (!on) bb = 0.0;
    double x = 0.0, r = bb, p = 0.0, v = 0.0;
    const double rh = r;
    const double rnorm0 = nb_max(fabs(r), red);
    double rnorm = rnorm0;
    int ret = 0, nit = 0;
    if (rnorm0 != 0.0) {
        double rho_1 = 0.0, alph = 0.0, omg = 0.0;
        for (nit = 1; nit <= maxiter; ++nit) {
            const double rho = nb_sum(rh * r, red);
            if (rho == 0.0) { ret = 1; break; }
            if (nit == 1) p = r;
            else {
                const double beta = (rho / rho_1) * (alph / omg);
                p = p - omg * v;
                p = r + beta * p;
            }
            v = apply(p);
            const double rhTv = nb_sum(rh * v, red);
            if (rhTv != 0.0) alph = rho / rhTv; else { ret = 2; break; }
            x = x + alph * p;
            const double sv = r - alph * v;
            rnorm = nb_max(fabs(sv), red);
            if (rnorm < eps_rel * rnorm0) break;
            const double t = apply(sv);
            const double tt = nb_sum(t * t, red), ts = nb_sum(t * sv, red);
            if (tt != 0.0) omg = ts / tt; else { ret = 3; break; }
            x = x + omg * sv;
            r = sv - omg * t;
            rnorm = nb_max(fabs(r), red);
            if (rnorm < eps_rel * rnorm0) break;
            if (omg == 0.0) { ret = 4; break; }
            rho_1 = rho;
        }
        if (ret == 0 && rnorm > eps_rel * rnorm0) ret = 8;
        if (!((ret == 0 || ret == 8) && rnorm < rnorm0)) x = 0.0;
    }
    if (tid == 0 && iters_out) atomicAdd(iters_out, nit);
    int ncalls = ret == 0 ? nub : nuf;
    if (ret != 0) { x = 0.0; ncalls += nuf; }
    __syncthreads();
    if (on) V[tid] = x;
    __syncthreads();
    for (int sw = 0; sw < ncalls * nsweeps; ++sw)
        for (int c = 0; c < 8; ++c) {
            const int cx = c & 1, cy = (c >> 1) & 1, cz = (c >> 2) & 1;
            if (on && (i & 1) == cx && (j & 1) == cy && (k & 1) == cz) {
                const double Ax = node_Ax_wrap(A, s, w, im, i, ip, jm, j, jp, km, k, kp, s0);
                x += (rhs0 - Ax) / s0;
                V[tid] = x;         // nodes of one colour are not neighbours of each other
            }
            __syncthreads();
        }
    if (on) cor(i, j, k) = x;
    __syncthreads();
    // periodic duplicates (index n) take the owner's value; ghosts are filled by the caller's FillBoundary
    const int m0 = n0 + 1, m1 = n1 + 1, m2 = n2 + 1;
    for (int idx = tid; idx < m0 * m1 * m2; idx += NBOT_NT) {
        const int qi = idx % m0, q = idx / m0;
        const int qj = q % m1, qk = q / m1;
        if (qi == n0 || qj == n1 || qk == n2)
            cor(lo0 + qi, lo1 + qj, lo2 + qk) = V[(qi == n0 ? 0 : qi) + n0 * ((qj == n1 ? 0 : qj) + n1 * (qk == n2 ? 0 : qk))];
    }
}

bool nodal_bottom_device_ok(const Geometry& g, const Layout& l)
{
    const bool enabled = tune("MG_DEVICE_BOTTOM", 1) != 0;
    if (!enabled || l.boxes.size() != 1) return false;      // global information only: every rank must build the same hierarchy
    const BoxD& b = l.boxes[0];
    long cells = 1;
    for (int d = 0; d < 3; ++d) {
        if (!g.periodic[d] || b.lo[d] != g.domain.lo[d] || b.hi[d] != g.domain.hi[d] || b.len(d) < 2) return false;
        cells *= b.len(d);
    }
    return cells <= NBOT_NT;
}

void nodal_bottom_solve(const Geometry& g, MultiFab& cor, const MultiFab& res, const MultiFab& sig, bool singular, double eps_rel, int maxiter,
                        int nsweeps, int nub, int nuf, int* d_iters)
{
    const Layout& l = *cor.layout;
    IAMRX_ASSERT(nodal_bottom_device_ok(g, l));
    if (l.nlocal() == 0) return;
    const BoxD& b = l.boxes[0];
    hipLaunchKernelGGL(k_nodal_bottom, dim3(1), dim3(NBOT_NT), 0, Context::get().stream, cor.d_tab, res.d_tab, sig.d_tab, make_w(g),
                       b.len(0), b.len(1), b.len(2), b.lo[0], b.lo[1], b.lo[2], singular ? 1 : 0, eps_rel, maxiter, nsweeps, nub, nuf, d_iters);
}

// General form of the device bottom solve: a single box of at most 8^3 cells whose directions are either periodic over the whole
// domain (unique nodes 0..n-1, index wrap) or bounded (nodes 0..n: Neumann walls by index reflection -- the ghost node -1 IS node 1,
// nodal_reflect_bc -- and Dirichlet / coarse-fine faces through the level's Dirichlet mask, whose nodes stay zero and never reach
// outside the box).  sigma is read with its ghost cells (periodic images / mirrored across walls, NodalMG::setSigma).  Weights of the
// dot products and of the mean: owner_weight's rule (1/2 per Neumann wall a node lies on; k_basic.hip).  Covers the levels of a
// refined patch and wall-bounded domains (LidDrivenCavity, RayleighTaylor); k_nodal_bottom stays the kernel of the periodic case.
constexpr int NBG_NT = 1024;
struct NBotGeom {
    int m[3];            // unknown nodes per direction
    int wrap[3];         // periodic over the box
    int lo[3], n[3];     // low cell index and number of cells of the box
    int half_lo[3], half_hi[3], dlo[3], dhi[3];     // Neumann walls of the domain and their node indices
};
__device__ __forceinline__ double nbg_sum(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < NBG_NT / 64; ++w) s += red[w];
    return s;
}
__device__ __forceinline__ double nbg_max(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < NBG_NT / 64; ++w) s = fmax(s, red[w]);
    return s;
}

__global__ void __launch_bounds__(NBG_NT) k_nodal_bottom_g(const FabD* __restrict__ cort, const FabD* __restrict__ rest, const FabD* __restrict__ sigt,
    const FabD* __restrict__ dmt, NodeW w, NBotGeom G, int singular, double mean_cnt, double eps_rel, int maxiter, int nsweeps, int nub, int nuf,
    int* __restrict__ iters_out)
{
    __shared__ double V[NBG_NT];
    __shared__ double red[NBG_NT / 64];
    const FabD cor = cort[0], res = rest[0], sg = sigt[0];
    const int m0 = G.m[0], m1 = G.m[1], m2 = G.m[2], M = m0 * m1 * m2, tid = threadIdx.x;
    const bool in = tid < M;
    const int ri = in ? tid % m0 : 0, rj = in ? (tid / m0) % m1 : 0, rk = in ? tid / (m0 * m1) : 0;
    const int i = G.lo[0] + ri, j = G.lo[1] + rj, k = G.lo[2] + rk;
    auto nb = [](int r, int m, int wrap, int dir) {     // relative index of the low / high neighbour: wrap or reflect
        if (dir < 0) return r == 0 ? (wrap ? m - 1 : 1) : r - 1;
        return r == m - 1 ? (wrap ? 0 : m - 2) : r + 1;
    };
    const int im = nb(ri, m0, G.wrap[0], -1), ip = nb(ri, m0, G.wrap[0], 1);
    const int jm = nb(rj, m1, G.wrap[1], -1), jp = nb(rj, m1, G.wrap[1], 1);
    const int km = nb(rk, m2, G.wrap[2], -1), kp = nb(rk, m2, G.wrap[2], 1);
    const bool masked = in && dmt != nullptr && dmt[0](i, j, k) != 0.0;
    const bool on = in && !masked;
    // weight of the node in sums and dot products
    double wt = on ? 1.0 : 0.0;
    {
        const int idx[3] = {i, j, k};
        for (int d = 0; d < 3; ++d) {
            if (G.wrap[d]) continue;
            if (idx[d] == G.lo[d] + G.n[d] && idx[d] != G.dhi[d] + 1) wt = 0.0;       // high face of the box inside the domain (coarse/fine): masked anyway
            if (G.half_lo[d] && idx[d] == G.dlo[d]) wt *= 0.5;
            if (G.half_hi[d] && idx[d] == G.dhi[d] + 1) wt *= 0.5;
        }
    }
    // sigma of the eight cells around the node (ghost cells included) and the diagonal
    double smmm = 0, spmm = 0, smpm = 0, sppm = 0, smmp = 0, spmp = 0, smpp = 0, sppp = 0;
    if (in) {
        smmm = sg(i - 1, j - 1, k - 1); spmm = sg(i, j - 1, k - 1); smpm = sg(i - 1, j, k - 1); sppm = sg(i, j, k - 1);
        smmp = sg(i - 1, j - 1, k); spmp = sg(i, j - 1, k); smpp = sg(i - 1, j, k); sppp = sg(i, j, k);
    }
    const double s0 = w.c * (smmm + spmm + smpm + sppm + smmp + spmp + smpp + sppp);
    auto X = [&](int a, int b, int c) { return V[a + m0 * (b + m1 * c)]; };
    auto Ax = [&]() -> double {          // node_Ax on the LDS vector (same expression as node_Ax_wrap)
        double y = X(ri, rj, rk) * s0;
        y += w.corner * (X(im, jm, km) * smmm + X(ip, jm, km) * spmm + X(im, jp, km) * smpm + X(ip, jp, km) * sppm
                       + X(im, jm, kp) * smmp + X(ip, jm, kp) * spmp + X(im, jp, kp) * smpp + X(ip, jp, kp) * sppp);
        y += w.ex * (X(ri, jm, km) * (smmm + spmm) + X(ri, jp, km) * (smpm + sppm) + X(ri, jm, kp) * (smmp + spmp) + X(ri, jp, kp) * (smpp + sppp));
        y += w.ey * (X(im, rj, km) * (smmm + smpm) + X(ip, rj, km) * (spmm + sppm) + X(im, rj, kp) * (smmp + smpp) + X(ip, rj, kp) * (spmp + sppp));
        y += w.ez * (X(im, jm, rk) * (smmm + smmp) + X(ip, jm, rk) * (spmm + spmp) + X(im, jp, rk) * (smpm + smpp) + X(ip, jp, rk) * (sppm + sppp));
        y += w.fx * (X(im, rj, rk) * (smmm + smpm + smmp + smpp) + X(ip, rj, rk) * (spmm + sppm + spmp + sppp));
        y += w.fy * (X(ri, jm, rk) * (smmm + spmm + smmp + spmp) + X(ri, jp, rk) * (smpm + sppm + smpp + sppp));
        y += w.fz * (X(ri, rj, km) * (smmm + spmm + smpm + sppm) + X(ri, rj, kp) * (smmp + spmp + smpp + sppp));
        return y;
    };
    auto apply = [&](double xv) -> double {
        __syncthreads();
        if (in) V[tid] = xv;
        __syncthreads();
        return on ? Ax() : 0.0;
    };
    const double rhs0 = on ? (double)res(i, j, k) : 0.0;
    double bb = rhs0;
    if (singular) bb -= nbg_sum(wt * rhs0, red) / mean_cnt;
    if (!on) bb = 0.0;
    double x = 0.0, r = bb, p = 0.0, v = 0.0;
    const double rh = r;
    const double rnorm0 = nbg_max(fabs(r), red);
    double rnorm = rnorm0;
    int ret = 0, nit = 0;
    if (rnorm0 != 0.0) {
        double rho_1 = 0.0, alph = 0.0, omg = 0.0;
        for (nit = 1; nit <= maxiter; ++nit) {
            const double rho = nbg_sum(wt * rh * r, red);
            if (rho == 0.0) { ret = 1; break; }
            if (nit == 1) p = r;
            else {
                const double beta = (rho / rho_1) * (alph / omg);
                p = p - omg * v;
                p = r + beta * p;
            }
            v = apply(p);
            const double rhTv = nbg_sum(wt * rh * v, red);
            if (rhTv != 0.0) alph = rho / rhTv; else { ret = 2; break; }
            x = x + alph * p;
            const double sv = r - alph * v;
            rnorm = nbg_max(fabs(sv), red);
            if (rnorm < eps_rel * rnorm0) break;
            const double t = apply(sv);
            const double tt = nbg_sum(wt * t * t, red), ts = nbg_sum(wt * t * sv, red);
            if (tt != 0.0) omg = ts / tt; else { ret = 3; break; }
            x = x + omg * sv;
            r = sv - omg * t;
            rnorm = nbg_max(fabs(r), red);
            if (rnorm < eps_rel * rnorm0) break;
            if (omg == 0.0) { ret = 4; break; }
            rho_1 = rho;
        }
        if (ret == 0 && rnorm > eps_rel * rnorm0) ret = 8;
        if (!((ret == 0 || ret == 8) && rnorm < rnorm0)) x = 0.0;
    }
    if (tid == 0 && iters_out) atomicAdd(iters_out, nit);
    int ncalls = ret == 0 ? nub : nuf;
    if (ret != 0) { x = 0.0; ncalls += nuf; }
    __syncthreads();
    if (in) V[tid] = x;
    __syncthreads();
    for (int sw = 0; sw < ncalls * nsweeps; ++sw)
        for (int c = 0; c < 8; ++c) {
            const int cx = c & 1, cy = (c >> 1) & 1, cz = (c >> 2) & 1;
            if (on && (i & 1) == cx && (j & 1) == cy && (k & 1) == cz) {
                x += (rhs0 - Ax()) / s0;
                V[tid] = x;
            }
            __syncthreads();
        }
    // all nodes of the box, the periodic duplicates (index n of a wrapped direction) included
    const int f0 = G.n[0] + 1, f1 = G.n[1] + 1, f2 = G.n[2] + 1;
    for (int q = tid; q < f0 * f1 * f2; q += NBG_NT) {
        const int qi = q % f0, qq = q / f0, qj = qq % f1, qk = qq / f1;
        const int a = (G.wrap[0] && qi == G.n[0]) ? 0 : qi, b = (G.wrap[1] && qj == G.n[1]) ? 0 : qj, c = (G.wrap[2] && qk == G.n[2]) ? 0 : qk;
        cor(G.lo[0] + qi, G.lo[1] + qj, G.lo[2] + qk) = V[a + m0 * (b + m1 * c)];
    }
}

static bool nbg_geom(const Geometry& g, const Layout& l, NBotGeom& G)
{
    if (l.boxes.size() != 1) return false;                      // global information only: every rank must build the same hierarchy
    const BoxD& b = l.boxes[0];
    long M = 1;
    for (int d = 0; d < 3; ++d) {
        if (b.len(d) < 2 || b.len(d) > 8) return false;
        G.wrap[d] = (g.periodic[d] && b.lo[d] == g.domain.lo[d] && b.hi[d] == g.domain.hi[d]) ? 1 : 0;
        G.m[d] = b.len(d) + (G.wrap[d] ? 0 : 1);
        G.lo[d] = b.lo[d]; G.n[d] = b.len(d);
        G.half_lo[d] = g.half_lo[d]; G.half_hi[d] = g.half_hi[d]; G.dlo[d] = g.domain.lo[d]; G.dhi[d] = g.domain.hi[d];
        M *= G.m[d];
    }
    return M <= NBG_NT;
}

bool nodal_bottom_device_ok_general(const Geometry& g, const Layout& l)
{
    const bool enabled = tune("MG_DEVICE_BOTTOM", 1) != 0 && tune("MG_DEVICE_BOTTOM_GENERAL", 1) != 0;
    NBotGeom G;
    return enabled && nbg_geom(g, l, G);
}

void nodal_bottom_solve_general(const Geometry& g, MultiFab& cor, const MultiFab& res, const MultiFab& sig, const MultiFab* dmask, bool singular,
                                double eps_rel, int maxiter, int nsweeps, int nub, int nuf, int* d_iters)
{
    const Layout& l = *cor.layout;
    NBotGeom G;
    IAMRX_ASSERT(nbg_geom(g, l, G) && sig.ngrow >= 1);
    if (l.nlocal() == 0) return;
    double cnt = 1.0;
    for (int d = 0; d < 3; ++d) cnt *= (double)g.domain.len(d);
    hipLaunchKernelGGL(k_nodal_bottom_g, dim3(1), dim3(NBG_NT), 0, Context::get().stream, cor.d_tab, res.d_tab, sig.d_tab,
                       dmask ? dmask->d_tab : nullptr, make_w(g), G, singular ? 1 : 0, cnt, eps_rel, maxiter, nsweeps, nub, nuf, d_iters);
}

// weighted Jacobi: x_new = x + (2/3) (rhs - A x)/s0 ; tmp holds x_new, then copied back by the caller
void nodal_jacobi(const Geometry& g, MultiFab& xnew, const MultiFab& x, const MultiFab& rhs, const MultiFab& sig, const MultiFab* dmask)
{
    if (x.nlocal() == 0) return;
    const NodeW w = make_w(g);
    const FabD *nt = xnew.d_tab, *xt = x.d_tab, *rt = rhs.d_tab, *st = sig.d_tab, *dt = dmask ? dmask->d_tab : nullptr;
    for_each(*x.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        if (dt && dt[f](i, j, k) != 0.0) { nt[f](i, j, k) = xt[f](i, j, k); return; }
        double s0;
        const double Ax = node_Ax(xt[f], st[f], w, i, j, k, s0);
        nt[f](i, j, k) = xt[f](i, j, k) + (2. / 3.) * (rt[f](i, j, k) - Ax) / s0;
    });
}

// mf = 0 on Dirichlet nodes (dmask != 0): residuals, operator rows, restricted residuals and prolonged corrections there
void nodal_zero_masked(MultiFab& mf, const MultiFab& dmask)
{
    if (mf.nlocal() == 0) return;
    const FabD *mt = mf.d_tab, *dt = dmask.d_tab;
    for_each(*mf.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        if (dt[f](i, j, k) != 0.0) mt[f](i, j, k) = 0.0;
    });
}

// Dirichlet node mask of a level (MLNodeLaplacian dirichlet mask): 1 on nodes with at least one of the 8 surrounding cells
// outside the problem -- beyond a Dirichlet (outflow) domain face or not covered by the level's boxes; cells beyond a Neumann
// wall mirror the cells inside, periodic images count.  cov: cell array, 1 ghost cell, 1 on the level's cells (ghosts filled
// from neighbours / periodic images, 0 elsewhere).  Ghost nodes of dm keep the value they had (the caller presets 1).
void nodal_build_dmask(const Geometry& g, MultiFab& dm, const MultiFab& cov, const DomainBC& bc)
{
    if (dm.nlocal() == 0) return;
    const FabD *dt = dm.d_tab, *ct = cov.d_tab;
    int lo[3], hi[3], tlo[3], thi[3];
    for (int d = 0; d < 3; ++d) {
        lo[d] = g.domain.lo[d]; hi[d] = g.domain.hi[d];
        tlo[d] = g.periodic[d] ? 0 : bc.lo[d]; thi[d] = g.periodic[d] ? 0 : bc.hi[d];
    }
    const int l0 = lo[0], l1 = lo[1], l2 = lo[2], h0 = hi[0], h1 = hi[1], h2 = hi[2];
    const int a0 = tlo[0], a1 = tlo[1], a2 = tlo[2], b0 = thi[0], b1 = thi[1], b2 = thi[2];
    for_each(*dm.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD c = ct[f];
        bool in = true;
        for (int q = 0; q < 8; ++q) {
            int ci = i - 1 + (q & 1), cj = j - 1 + ((q >> 1) & 1), ck = k - 1 + ((q >> 2) & 1);
            bool out = false;
            if (ci < l0) { if (a0 == lo_neumann) ci = 2 * l0 - 1 - ci; else if (a0 != 0) out = true; }
            else if (ci > h0) { if (b0 == lo_neumann) ci = 2 * h0 + 1 - ci; else if (b0 != 0) out = true; }
            if (cj < l1) { if (a1 == lo_neumann) cj = 2 * l1 - 1 - cj; else if (a1 != 0) out = true; }
            else if (cj > h1) { if (b1 == lo_neumann) cj = 2 * h1 + 1 - cj; else if (b1 != 0) out = true; }
            if (ck < l2) { if (a2 == lo_neumann) ck = 2 * l2 - 1 - ck; else if (a2 != 0) out = true; }
            else if (ck > h2) { if (b2 == lo_neumann) ck = 2 * h2 + 1 - ck; else if (b2 != 0) out = true; }
            if (out || c(ci, cj, ck) == 0.0) in = false;
        }
        dt[f](i, j, k) = in ? 0.0 : 1.0;
    });
}

// full weighting (1,2,1)^3/64; the fine array needs one filled ghost-node layer
void nodal_restrict(MultiFab& crse, const MultiFab& fine)
{
    if (crse.nlocal() == 0) return;
    const FabD *ct = crse.d_tab, *ft = fine.d_tab;
    for_each(*crse.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD fa = ft[f];
        const int ii = 2 * i, jj = 2 * j, kk = 2 * k;
        double s = 0.0;
        for (int dk = -1; dk <= 1; ++dk)
            for (int dj = -1; dj <= 1; ++dj)
                for (int di = -1; di <= 1; ++di) {
                    const double w = (di == 0 ? 2. : 1.) * (dj == 0 ? 2. : 1.) * (dk == 0 ? 2. : 1.);
                    s += w * fa(ii + di, jj + dj, kk + dk);
                }
        ct[f](i, j, k) = s * (1. / 64.);
    });
}

#ifndef IAMRX_INTERP_EXP
#define IAMRX_INTERP_EXP 0      // timing experiments on k_nodal_interp_lds (wrong results): 1: constant side weights (no sigma loads), 2: plain store instead of the read-modify-write, 4: no divisions
#endif
// sigma-weighted interpolation (mlndlap_interpadd_aa)
__device__ __forceinline__ double w_side(const FabD& s, int i, int j, int k, int d, int side)
{
    int lo[3] = {i - 1, j - 1, k - 1}, hi[3] = {i, j, k};
    lo[d] = hi[d] = (d == 0 ? i : (d == 1 ? j : k)) - 1 + side;
    double r = 0.0;
    for (int c = lo[2]; c <= hi[2]; ++c) for (int b = lo[1]; b <= hi[1]; ++b) for (int a = lo[0]; a <= hi[0]; ++a) r += s(a, b, c);
    return r;
}
__device__ __forceinline__ double interp_line(const FabD& c, const FabD& s, int i, int j, int k, int ic, int jc, int kc, int d)
{
    const double w1 = w_side(s, i, j, k, d, 0), w2 = w_side(s, i, j, k, d, 1);
    const double c2 = c(ic + (d == 0), jc + (d == 1), kc + (d == 2));
    return (c(ic, jc, kc) * w1 + c2 * w2) / (w1 + w2);
}
__device__ __forceinline__ double interp_face(const FabD& c, const FabD& s, int i, int j, int k, int ic, int jc, int kc, int d1, int d2)
{
    const double w1 = w_side(s, i, j, k, d1, 0), w2 = w_side(s, i, j, k, d1, 1), w3 = w_side(s, i, j, k, d2, 0), w4 = w_side(s, i, j, k, d2, 1);
    const int e1[3] = {d1 == 0, d1 == 1, d1 == 2}, e2[3] = {d2 == 0, d2 == 1, d2 == 2};
    double r = 0.0;
    r += w1 * interp_line(c, s, i - e1[0], j - e1[1], k - e1[2], ic, jc, kc, d2);
    r += w2 * interp_line(c, s, i + e1[0], j + e1[1], k + e1[2], ic + e1[0], jc + e1[1], kc + e1[2], d2);
    r += w3 * interp_line(c, s, i - e2[0], j - e2[1], k - e2[2], ic, jc, kc, d1);
    r += w4 * interp_line(c, s, i + e2[0], j + e2[1], k + e2[2], ic + e2[0], jc + e2[1], kc + e2[2], d1);
    return r / (w1 + w2 + w3 + w4);
}

// the six side weights of a fine node from the 8 cells around it, summed in w_side's order
template <class SA>
__device__ __forceinline__ void side_weights(const SA& s, int i, int j, int k, double w[6])
{
#if IAMRX_INTERP_EXP & 1
    for (int t = 0; t < 6; ++t) w[t] = 1.0 + 1e-3 * t;
    return;
#endif
    double g[2][2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a) g[c][b][a] = s(i - 1 + a, j - 1 + b, k - 1 + c);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        w[t] = ((g[0][0][t] + g[0][1][t]) + g[1][0][t]) + g[1][1][t];
        w[2 + t] = ((g[0][t][0] + g[0][t][1]) + g[1][t][0]) + g[1][t][1];
        w[4 + t] = ((g[t][0][0] + g[t][0][1]) + g[t][1][0]) + g[t][1][1];
    }
}

// Staged version of the interpolation below: one workgroup owns the fine nodes of CX x CY x CZ coarse cells.  The recursion of
// mlndlap_interpadd_aa (edge nodes from coarse nodes, face nodes from edge nodes, centre nodes from face nodes) runs class by class
// on the (2C+1)^3 fine-node tile in LDS, so that every intermediate value is computed once (the per-coarse-node version recomputes
// the edge values inside the face values inside the centre value: ~200 dependent sigma loads for one centre node), then the tile is
// added to the fine array with unit-stride stores.  Same expressions, same association order: bit-identical results.
// the fine nodes of one parity class (PX, PY, PZ) of the tile, enumerated densely (all lanes busy)
template <int PX, int PY, int PZ, int CX, int CY, int CZ, class SA>
__device__ __forceinline__ void interp_class(double (*T)[2 * CY + 1][2 * CX + 1], const FabD& c, const SA& s, int fi0, int fj0, int fk0,
                                             int nhi0, int nhi1, int nhi2, int tid)
{
    constexpr int NX = PX ? CX : CX + 1, NY = PY ? CY : CY + 1, NZ = PZ ? CZ : CZ + 1, CLS = PX + PY + PZ;
    for (int idx = tid; idx < NX * NY * NZ; idx += 256) {
        const int lx = 2 * (idx % NX) + PX, q = idx / NX, ly = 2 * (q % NY) + PY, lz = 2 * (q / NY) + PZ;
        const int i = fi0 + lx, j = fj0 + ly, k = fk0 + lz;
        if (i > nhi0 || j > nhi1 || k > nhi2) continue;
        if (CLS == 0) { T[lz][ly][lx] = c(i >> 1, j >> 1, k >> 1); continue; }
        double w[6];
        side_weights(s, i, j, k, w);
        if (CLS == 1) {
            constexpr int d = PX ? 0 : (PY ? 1 : 2);
            const double w1 = w[2 * d], w2 = w[2 * d + 1];
            T[lz][ly][lx] = (IAMRX_INTERP_EXP & 4) ? (T[lz - PZ][ly - PY][lx - PX] * w1 + T[lz + PZ][ly + PY][lx + PX] * w2) * (w1 + w2) : (T[lz - PZ][ly - PY][lx - PX] * w1 + T[lz + PZ][ly + PY][lx + PX] * w2) / (w1 + w2);
        } else if (CLS == 2) {
            constexpr int d1 = PX ? 0 : 1, d2 = PZ ? 2 : 1;
            constexpr int ax = d1 == 0, ay = d1 == 1, by = d2 == 1, bz = d2 == 2;
            const double w1 = w[2 * d1], w2 = w[2 * d1 + 1], w3 = w[2 * d2], w4 = w[2 * d2 + 1];
            double r = 0.0;
            r += w1 * T[lz][ly - ay][lx - ax];
            r += w2 * T[lz][ly + ay][lx + ax];
            r += w3 * T[lz - bz][ly - by][lx];
            r += w4 * T[lz + bz][ly + by][lx];
            T[lz][ly][lx] = (IAMRX_INTERP_EXP & 4) ? r * (w1 + w2 + w3 + w4) : r / (w1 + w2 + w3 + w4);
        } else {
            T[lz][ly][lx] = (w[0] * T[lz][ly][lx - 1] + w[1] * T[lz][ly][lx + 1] + w[2] * T[lz][ly - 1][lx] + w[3] * T[lz][ly + 1][lx]
                             + w[4] * T[lz - 1][ly][lx] + w[5] * T[lz + 1][ly][lx]) / (w[0] + w[1] + w[2] + w[3] + w[4] + w[5]);
        }
    }
}

template <int CX, int CY, int CZ>
__global__ void __launch_bounds__(256) k_nodal_interp_lds(const BoxD* __restrict__ fboxes, const FabD* __restrict__ ft, const FabD* __restrict__ ct,
    const FabD* __restrict__ st, int ntx, int nty)
{
    constexpr int FX = 2 * CX + 1, FY = 2 * CY + 1, FZ = 2 * CZ + 1, NF = FX * FY * FZ;
    __shared__ double T[FZ][FY][FX];
    const int fab = blockIdx.y;
    const BoxD vb = fboxes[fab];
    const int bid = blockIdx.x;
    const int tix = bid % ntx, r1 = bid / ntx, tiy = r1 % nty, tiz = r1 / nty;
    const int nhi0 = vb.hi[0] + 1, nhi1 = vb.hi[1] + 1, nhi2 = vb.hi[2] + 1;
    const int fi0 = vb.lo[0] + tix * 2 * CX, fj0 = vb.lo[1] + tiy * 2 * CY, fk0 = vb.lo[2] + tiz * 2 * CZ;
    if (fi0 >= nhi0 || fj0 >= nhi1 || fk0 >= nhi2) return;
    const FabD c = ct[fab], sg = st[fab], fa = ft[fab];
    const int tid = threadIdx.x;
    // (sigma of the tile in LDS -- one unit-stride pass, parity-split rows, no bank conflicts -- was measured twice: 230 / 235 us against 177 us.
    // The side weights' 8 sigma loads per node and class ARE 103 of the 177 us (constant weights: 74 us, tools/r5_interp_exp.sh), but 27 KB more
    // LDS per workgroup take the occupancy from 7 to 3 workgroups per CU, and the phases between the barriers are too short to do without it.
    // What would work is one coarse cell per thread with its 27 sigma values in registers for all four phases -- not built.)
    const FabD& s = sg;
#define IAMRX_ICLS(PX, PY, PZ) interp_class<PX, PY, PZ, CX, CY, CZ>(T, c, s, fi0, fj0, fk0, nhi0, nhi1, nhi2, tid)
    IAMRX_ICLS(0, 0, 0);
    __syncthreads();
    IAMRX_ICLS(1, 0, 0); IAMRX_ICLS(0, 1, 0); IAMRX_ICLS(0, 0, 1);
    __syncthreads();
    IAMRX_ICLS(1, 1, 0); IAMRX_ICLS(1, 0, 1); IAMRX_ICLS(0, 1, 1);
    __syncthreads();
    IAMRX_ICLS(1, 1, 1);
    __syncthreads();
#undef IAMRX_ICLS
    for (int idx = tid; idx < NF; idx += 256) {
        const int lx = idx % FX, q = idx / FX, ly = q % FY, lz = q / FY;
        const int i = fi0 + lx, j = fj0 + ly, k = fk0 + lz;
        if (i > nhi0 || j > nhi1 || k > nhi2) continue;
        if ((lx == 2 * CX && i != nhi0) || (ly == 2 * CY && j != nhi1) || (lz == 2 * CZ && k != nhi2)) continue;   // the next tile owns it
        if (IAMRX_INTERP_EXP & 2) fa(i, j, k) = T[lz][ly][lx]; else fa(i, j, k) += T[lz][ly][lx];
    }
}

static bool nodal_interp_lds_enabled()
{
    const bool on = tune("NODAL_INTERP_LDS", 1) != 0;
    return on;
}

// one thread per COARSE node: it produces the (up to) 8 fine nodes 2*(ic,jc,kc) + {0,1}^3, so that every lane of a wavefront
// walks the same sequence of node classes (coincident, 3 edge, 3 face, 1 centre class) instead of diverging 8 ways
void nodal_interp_add(MultiFab& fine, const MultiFab& crse, const MultiFab& sig_fine)
{
    if (fine.nlocal() == 0) return;
    if (nodal_interp_lds_enabled()) {
        const Layout& l = *fine.layout;
        bool even = true;                                   // fine boxes start on even indices (they are refinements of the coarse boxes)
        for (int d = 0; d < 3; ++d) even = even && l.all_lo_even[d];
        if (even) {
            constexpr int CX = 16, CY = 4, CZ = 4;
            const int ntx = (l.max_len[0] + 2 * CX - 1) / (2 * CX), nty = (l.max_len[1] + 2 * CY - 1) / (2 * CY), ntz = (l.max_len[2] + 2 * CZ - 1) / (2 * CZ);
            dim3 grid((unsigned)(ntx * nty * ntz), (unsigned)l.nlocal());
            hipLaunchKernelGGL((k_nodal_interp_lds<CX, CY, CZ>), grid, dim3(256), 0, Context::get().stream, l.d_boxes, fine.d_tab, crse.d_tab, sig_fine.d_tab, ntx, nty);
            return;
        }
    }
    const FabD *ft = fine.d_tab, *ct = crse.d_tab, *st = sig_fine.d_tab;
    const BoxD* fb = fine.layout->d_boxes;
    for_each(*crse.layout, node_type(), 0, Context::get().stream, [=] __device__(int ic, int jc, int kc, int f) {
        const FabD c = ct[f], s = st[f], fa = ft[f];
        const BoxD vb = fb[f];
        const int nhi0 = vb.hi[0] + 1, nhi1 = vb.hi[1] + 1, nhi2 = vb.hi[2] + 1;
        const int i0 = 2 * ic, j0 = 2 * jc, k0 = 2 * kc;
        const bool xi = i0 + 1 <= nhi0, xj = j0 + 1 <= nhi1, xk = k0 + 1 <= nhi2;
        fa(i0, j0, k0) += c(ic, jc, kc);
        if (xi) fa(i0 + 1, j0, k0) += interp_line(c, s, i0 + 1, j0, k0, ic, jc, kc, 0);
        if (xj) fa(i0, j0 + 1, k0) += interp_line(c, s, i0, j0 + 1, k0, ic, jc, kc, 1);
        if (xk) fa(i0, j0, k0 + 1) += interp_line(c, s, i0, j0, k0 + 1, ic, jc, kc, 2);
        if (xi && xj) fa(i0 + 1, j0 + 1, k0) += interp_face(c, s, i0 + 1, j0 + 1, k0, ic, jc, kc, 0, 1);
        if (xi && xk) fa(i0 + 1, j0, k0 + 1) += interp_face(c, s, i0 + 1, j0, k0 + 1, ic, jc, kc, 0, 2);
        if (xj && xk) fa(i0, j0 + 1, k0 + 1) += interp_face(c, s, i0, j0 + 1, k0 + 1, ic, jc, kc, 1, 2);
        if (xi && xj && xk) {
            const int i = i0 + 1, j = j0 + 1, k = k0 + 1;
            double w[6];
            for (int d = 0; d < 3; ++d) { w[2 * d] = w_side(s, i, j, k, d, 0); w[2 * d + 1] = w_side(s, i, j, k, d, 1); }
            const double v = (w[0] * interp_face(c, s, i - 1, j, k, ic, jc, kc, 1, 2) + w[1] * interp_face(c, s, i + 1, j, k, ic + 1, jc, kc, 1, 2)
               + w[2] * interp_face(c, s, i, j - 1, k, ic, jc, kc, 0, 2) + w[3] * interp_face(c, s, i, j + 1, k, ic, jc + 1, kc, 0, 2)
               + w[4] * interp_face(c, s, i, j, k - 1, ic, jc, kc, 0, 1) + w[5] * interp_face(c, s, i, j, k + 1, ic, jc, kc + 1, 0, 1))
              / (w[0] + w[1] + w[2] + w[3] + w[4] + w[5]);
            fa(i, j, k) += v;
        }
    });
}

// rhs(node) = FE divergence of the cell-centred velocity (mlndlap_divu); vel needs 1 filled ghost cell
// bc (may be null = periodic / interior only): cells outside a Neumann wall contribute zero velocity and the rhs of
// wall nodes is doubled per wall direction (mlndlap_divu + mlndlap_impose_neumann_bc)
void nodal_divu(const Geometry& g, MultiFab& rhs, const MultiFab& vel, int vcomp, const DomainBC* bc)
{
    if (rhs.nlocal() == 0) return;
    const FabD *rt = rhs.d_tab, *vt = vel.d_tab;
    const double fx = 0.25 / g.dx[0], fy = 0.25 / g.dx[1], fz = 0.25 / g.dx[2];
    // set_boundary_velocity (Source/Projection.cpp:2570-2663) + mlndlap_divu: cells outside a Neumann wall carry no velocity,
    // outside an inflow face only the normal component (the inflow value) survives; the rhs of wall / inflow nodes is doubled
    int wl[3], wh[3], il[3], ih[3];     // Neumann wall flags, inflow flags
    for (int d = 0; d < 3; ++d) {
        wl[d] = (bc && !g.periodic[d] && bc->lo[d] == lo_neumann) ? 1 : 0;
        wh[d] = (bc && !g.periodic[d] && bc->hi[d] == lo_neumann) ? 1 : 0;
        il[d] = (bc && !g.periodic[d] && bc->lo[d] == lo_inflow) ? 1 : 0;
        ih[d] = (bc && !g.periodic[d] && bc->hi[d] == lo_inflow) ? 1 : 0;
    }
    const int wl0 = wl[0], wl1 = wl[1], wl2 = wl[2], wh0 = wh[0], wh1 = wh[1], wh2 = wh[2];
    const int il0 = il[0], il1 = il[1], il2 = il[2], ih0 = ih[0], ih1 = ih[1], ih2 = ih[2];
    const int dl0 = g.domain.lo[0], dl1 = g.domain.lo[1], dl2 = g.domain.lo[2], dh0 = g.domain.hi[0], dh1 = g.domain.hi[1], dh2 = g.domain.hi[2];
    for_each(*rhs.layout, node_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD v = vt[f];
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
            const int ci = i - 1 + cx, cj = j - 1 + cy, ck = k - 1 + cz;
            const bool outside = (wl0 && ci < dl0) || (wh0 && ci > dh0) || (wl1 && cj < dl1) || (wh1 && cj > dh1) || (wl2 && ck < dl2) || (wh2 && ck > dh2);
            const bool in0 = (il0 && ci < dl0) || (ih0 && ci > dh0), in1 = (il1 && cj < dl1) || (ih1 && cj > dh1), in2 = (il2 && ck < dl2) || (ih2 && ck > dh2);
            sx += (cx ? 1.0 : -1.0) * ((outside || in1 || in2) ? 0.0 : v(ci, cj, ck, vcomp));
            sy += (cy ? 1.0 : -1.0) * ((outside || in0 || in2) ? 0.0 : v(ci, cj, ck, vcomp + 1));
            sz += (cz ? 1.0 : -1.0) * ((outside || in0 || in1) ? 0.0 : v(ci, cj, ck, vcomp + 2));
        }
        double r = 0.0;
        r += fx * sx; r += fy * sy; r += fz * sz;
        if ((wl0 || il0) && i == dl0) r *= 2.0;
        if ((wh0 || ih0) && i == dh0 + 1) r *= 2.0;
        if ((wl1 || il1) && j == dl1) r *= 2.0;
        if ((wh1 || ih1) && j == dh1 + 1) r *= 2.0;
        if ((wl2 || il2) && k == dl2) r *= 2.0;
        if ((wh2 || ih2) && k == dh2 + 1) r *= 2.0;
        rt[f](i, j, k) = r;
    });
}

// vel -= sig * grad(phi) (mlndlap_mknewu_aa); gp (optional) = grad(phi) stored or accumulated (compGrad)
void nodal_mknewu(const Geometry& g, MultiFab* vel, int vcomp, const MultiFab& phi, const MultiFab* sig, MultiFab* gp, bool gp_increment)
{
    if (phi.nlocal() == 0) return;
    const FabD* pt = phi.d_tab;
    const FabD* vt = vel ? vel->d_tab : nullptr;
    const FabD* st = sig ? sig->d_tab : nullptr;
    const FabD* gt = gp ? gp->d_tab : nullptr;
    const double fac[3] = {0.25 / g.dx[0], 0.25 / g.dx[1], 0.25 / g.dx[2]};
    const double f0 = fac[0], f1 = fac[1], f2 = fac[2];
    for_each(*phi.layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD p = pt[f];
        double s[3] = {0.0, 0.0, 0.0};
        for (int nz = 0; nz < 2; ++nz) for (int ny = 0; ny < 2; ++ny) for (int nx = 0; nx < 2; ++nx) {
            const double v = p(i + nx, j + ny, k + nz);
            s[0] += (nx ? 1.0 : -1.0) * v;
            s[1] += (ny ? 1.0 : -1.0) * v;
            s[2] += (nz ? 1.0 : -1.0) * v;
        }
        const double gr[3] = {f0 * s[0], f1 * s[1], f2 * s[2]};
        if (vt) {
            const double sg = st[f](i, j, k);
            const double fc[3] = {f0, f1, f2};
            for (int d = 0; d < 3; ++d) vt[f](i, j, k, vcomp + d) -= sg * fc[d] * s[d];
        }
        if (gt) for (int d = 0; d < 3; ++d) { if (gp_increment) gt[f](i, j, k, d) += gr[d]; else gt[f](i, j, k, d) = gr[d]; }
    });
}

}  // namespace iamrx
