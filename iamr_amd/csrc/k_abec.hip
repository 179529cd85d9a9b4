// iamr_amd/csrc/k_abec.hip -- hand-written gfx950 kernels for the cell-centred (alpha*a - beta div b grad)
// operator: red-black Gauss-Seidel, residual/apply, restriction, prolongation, face coefficient
// coarsening, fluxes, and the MAC-projection helpers.
//
// Role: AMReX MLABecLaplacian / MLCellLinOp primitives used through Hydro::MacProjector at reference
// Source/MacProj.cpp:1150-1183 and through MLMG at Source/Diffusion.cpp:327-345.
// All kernels are HBM-bandwidth bound (7-point stencil, fp64): rows are read as contiguous 64-lane
// wavefront segments, each workgroup marches TZ planes so the k-1/k/k+1 planes are re-used from L1/L2.
#include "kernels.h"
#include "launch.h"
#include <map>
#include <array>
#include <cstring>
#include <vector>
#include <algorithm>

namespace iamrx {

struct CfC1 { double c1[3][3]; double c2[3][3], c3[3][3]; int maxorder; int maintain; };   // CfTab::c[d][NX-2][1..3]; maintain: see cf_maintain

// Coarse/fine ghost cells kept current by the colour passes themselves (homogeneous data, one component).  The ghost G of a line normal
// to a coarse/fine face is c1 p1 + c2 p2 + c3 p3 of the first NX - 1 cells inside (k_cf_fill).  A pass of colour C reads G only in
// lines whose first cell p1 has colour C; p1 and p3 then have colour C, p2 the other one, so the G such a pass reads went stale in the
// pass of the OTHER colour, through p2 alone.  The thread that updates a cell at distance 2 from a coarse/fine face therefore rewrites
// that line's G with its new value (p1, p3 do not change during its pass; nobody reads this G during its pass): the next pass finds
// exactly what a k_cf_fill launch in front of it would have written (same expression, same operands) -- one fill in front of the first
// pass of a smoothing call instead of one per pass.  NX == 2 (boxes one cell wide): G = c1 p1, rewritten by the thread of p1.
__device__ __forceinline__ void cf_maintain(const FabD& phi, const FabD& cfm, const BoxD& b, const CfC1& t, int i, int j, int k, double pnew)
{
    const int idx[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) {
        const int NX = min(b.len(d) + 1, t.maxorder);
        if (NX < 2) continue;
        const int target = NX == 2 ? 1 : 2;
        for (int side = 0; side < 2; ++side) {
            const int s = side == 0 ? 1 : -1, f = side == 0 ? b.lo[d] : b.hi[d];
            if ((idx[d] - f) * s + 1 != target) continue;
            int g[3] = {i, j, k};
            g[d] = f - s;
            if (cfm(g[0], g[1], g[2]) != 1.0) continue;
            const double cq[4] = {0.0, t.c1[d][NX - 2], t.c2[d][NX - 2], t.c3[d][NX - 2]};
            double v = 0.0;
            int m[3] = {i, j, k};
            for (int q = 1; q < NX; ++q) {
                m[d] = g[d] + q * s;
                v += (q == target ? pnew : (double)phi(m[0], m[1], m[2], 0)) * cq[q];
            }
            phi(g[0], g[1], g[2], 0) = v;
        }
    }
}

// cf_maintain for a thread that still holds the six neighbours of its cell (nb = {x-, x+, y-, y+, z-, z+}, values read before the update: they
// have the other colour and do not change during the pass): the cells the ghost formula needs besides the updated one ARE those neighbours,
// so the rewrite costs a mask load and a store instead of the loop of dependent loads -- in the lean kernels every wavefront holds a lane
// next to an x face and used to run that loop once per plane (190 instead of 126 us per 256^3 colour pass on a refined level).  Same sum,
// same order as cf_maintain.
// per box: NX and the three weights of each direction, picked from the level's table once per thread
struct CfDir { int nx[3]; double c1[3], c2[3], c3[3]; };
__device__ __forceinline__ CfDir cf_dir(const BoxD& b, const CfC1& t)
{
    CfDir r;
    for (int d = 0; d < 3; ++d) {
        const int NX = min(b.len(d) + 1, t.maxorder);
        r.nx[d] = NX;
        const int q = NX >= 2 ? NX - 2 : 0;
        r.c1[d] = t.c1[d][q]; r.c2[d] = t.c2[d][q]; r.c3[d] = t.c3[d][q];
    }
    return r;
}
__device__ __forceinline__ void cf_maintain_nb(const FabD& phi, const FabD& cfm, const BoxD& b, const CfDir& t, int i, int j, int k, double pnew, const double nb[6])
{
    const int idx[3] = {i, j, k};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int NX = t.nx[d];
        if (NX < 2) continue;
        const int target = NX == 2 ? 1 : 2;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int s = side == 0 ? 1 : -1, f = side == 0 ? b.lo[d] : b.hi[d];
            if ((idx[d] - f) * s + 1 != target) continue;
            int g[3] = {i, j, k};
            g[d] = f - s;
            if (cfm(g[0], g[1], g[2]) != 1.0) continue;
            double v = 0.0;
            if (target == 1) v += pnew * t.c1[d];
            else {
                v += nb[2 * d + side] * t.c1[d];                 // the cell next to the face
                v += pnew * t.c2[d];
                if (NX > 3) v += nb[2 * d + 1 - side] * t.c3[d];  // the third cell of the line
            }
            phi(g[0], g[1], g[2], 0) = v;
        }
    }
}

struct GsrbBC {
    int dlo[3], dhi[3];
    double cflo[3][3], cfhi[3][3];   // [comp][dir]: coefficient of the first interior cell in the ghost formula; 0 for periodic
    int nbc;                         // 1: the same BC for every component
};

// Walls handled INSIDE the colour-pass kernels (k_abec_gsrb, k_abec_gsrb1) on a level that is one box spanning its domain: the value beyond a
// domain face is k_abec_bc's homogeneous ghost formula on the values at hand, g = p0 c1 (+ p_in c2: Dirichlet of order 3; p_in = the cell's
// neighbour on the inner side), so no ghost cell of phi is read in a non-periodic direction and the k_abec_bc launch in front of every colour
// pass of a smoothing call goes (LidDrivenCavity 256^3: ~770 launches of ~5.5 us per step on the coarse levels of the MAC, tensor and scalar
// solves).  A ghost fill between the colours sees the cell at the wall with its old value and the inner neighbour (other colour) current --
// exactly what the pass itself holds.  on = 0: the ghost cells are read (filled by the caller).  [comp][dir]
struct WallK { int on; int per[3]; double c1lo[3][3], c2lo[3][3], c1hi[3][3], c2hi[3][3]; };
__device__ __forceinline__ double wallk_ghost(double p0, double pin, double c1, double c2)
{
    double g = p0 * c1;
    if (c2 != 0.0) g += pin * c2;
    return g;
}
// Lagrange weights for ghost-cell extrapolation through a Dirichlet face (AMReX poly_interp_coeff):
// points x = {0 (face), 0.5, 1.5, 2.5}, evaluated at -0.5
static void poly_interp_coeff(double xi, const double* x, int N, double* c)
{
    for (int j = 0; j < N; ++j) {
        double num = 1.0, den = 1.0;
        for (int i = 0; i < N; ++i) { if (i == j) continue; num *= xi - x[i]; den *= x[j] - x[i]; }
        c[j] = num / den;
    }
}
static void dirichlet_coefs(int blen, int maxorder, double c[4], int& NX)
{
    NX = blen + 1 < maxorder ? blen + 1 : maxorder;
    const double x[4] = {0.0, 0.5, 1.5, 2.5};
    c[0] = c[1] = c[2] = c[3] = 0.0;
    if (NX >= 2) poly_interp_coeff(-0.5, x, NX, c);
}

static GsrbBC make_gsrb_bc(const Geometry& g, const DomainBC* bcs, int nbc)
{
    GsrbBC r;
    r.nbc = nbc;
    for (int d = 0; d < 3; ++d) { r.dlo[d] = g.domain.lo[d]; r.dhi[d] = g.domain.hi[d]; }
    for (int n = 0; n < 3; ++n) {
        const DomainBC& bc = bcs[n < nbc ? n : 0];
        for (int d = 0; d < 3; ++d) {
            r.cflo[n][d] = r.cfhi[n][d] = 0.0;
            if (g.periodic[d]) continue;
            for (int side = 0; side < 2; ++side) {
                const int b = side == 0 ? bc.lo[d] : bc.hi[d];
                double cf = 0.0;
                if (b == lo_neumann) cf = 1.0;
                else if (b == lo_reflect_odd) cf = -1.0;
                else if (b == lo_dirichlet) { double c[4]; int NX; dirichlet_coefs(g.domain.len(d), bc.maxorder, c, NX); cf = NX >= 2 ? c[1] : 0.0; }
                (side == 0 ? r.cflo[n][d] : r.cfhi[n][d]) = cf;
            }
        }
    }
    return r;
}

// ---------------------------------------------------------------------------- GSRB
// One thread per cell of the active colour: lane m of a row handles i = lo + 2m + parity, so a
// wavefront sweeps 128 consecutive cells of a row; phi(i+-1) and the b pairs are contiguous across
// lanes.  b arrays have either ncomp comps or 1 (broadcast).
// SHARE: several components on one shared 1-component coefficient set (eta form of the tensor operator)
// SIG: one component, the face coefficients recomputed from the cell-centred array they were made of (AbecCoef::sig; bxt = that array)
struct BUni { double v[3]; };
// BMODE 0: b arrays; 1 (SIG): recomputed from AbecCoef::sig; 2 (UNI): the constants of AbecCoef::bu
template <bool SHARE, int BMODE = 0>
__global__ void __launch_bounds__(256) k_abec_gsrb(Tiling t, const BoxD* __restrict__ boxes,
    const FabD* __restrict__ phit, const FabD* __restrict__ rhst, const FabD* __restrict__ at,
    const FabD* __restrict__ bxt, const FabD* __restrict__ byt, const FabD* __restrict__ bzt,
    double alpha, double dhx, double dhy, double dhz, int redblack, double omega, int ncomp, int bnc, GsrbBC bc, int shell_only, int tens, int wrap,
    const FabD* __restrict__ cfmt, CfC1 cfc, int sig_comp = 0, double sig_scale = 1.0, BUni bu = BUni(), const WallK* __restrict__ wkp = nullptr)
{
    constexpr bool SIG = BMODE == 1, UNI = BMODE == 2;
    const int fab = tile_fab(t);
    const BoxD b = boxes[fab];
    BoxD hb = b;
    hb.hi[0] = b.lo[0] + (b.len(0) + 1) / 2 - 1;
    int ih, j, k0, k1;
    if (!tile_ijk(t, hb, ih, j, k0, k1)) return;
    const FabD phi = phit[fab], rhs = rhst[fab], bX = bxt[fab], bY = byt[fab], bZ = bzt[fab];
    const bool has_a = (at != nullptr) && alpha != 0.0;
    FabD A; if (has_a) A = at[fab];
    // coarse/fine faces: the first-interior-cell weight of the ghost formula depends on the box length (NX = min(len + 1, maxorder))
    const bool cf = cfmt != nullptr;
    FabD cfm; if (cf) cfm = cfmt[fab];
    const double c1x = cf ? cfc.c1[0][min(b.len(0) + 1, cfc.maxorder) - 2] : 0.0, c1y = cf ? cfc.c1[1][min(b.len(1) + 1, cfc.maxorder) - 2] : 0.0;
    const double c1z = cf ? cfc.c1[2][min(b.len(2) + 1, cfc.maxorder) - 2] : 0.0;
    // wrap: the box spans a fully periodic domain, neighbours across the box faces are the periodic images inside the same box
    // (no ghost fill needed in front of the sweep)
    const int jm = (wrap && j == b.lo[1]) ? b.hi[1] : j - 1, jp = (wrap && j == b.hi[1]) ? b.lo[1] : j + 1;
    if (SHARE) {
    // plane loop outside, component loop inside: a one-component coefficient set (bnc == 1: scalar problems and the eta form of
    // the tensor operator) is then read once per cell and shared by all components
    for (int k = k0; k <= k1; ++k) {
        const int i = b.lo[0] + 2 * (ih - b.lo[0]) + ((b.lo[0] + j + k + redblack) & 1);
        if (i > b.hi[0]) continue;
        if (shell_only && i > b.lo[0] && i < b.hi[0] && j > b.lo[1] && j < b.hi[1] && k > b.lo[2] && k < b.hi[2]) continue;
        const int im = (wrap && i == b.lo[0]) ? b.hi[0] : i - 1, ip = (wrap && i == b.hi[0]) ? b.lo[0] : i + 1;
        const int km = (wrap && k == b.lo[2]) ? b.hi[2] : k - 1, kp = (wrap && k == b.hi[2]) ? b.lo[2] : k + 1;
        double b1xm = 0, b1xp = 0, b1ym = 0, b1yp = 0, b1zm = 0, b1zp = 0;
        if (UNI) { b1xm = b1xp = bu.v[0]; b1ym = b1yp = bu.v[1]; b1zm = b1zp = bu.v[2]; }
        else if (bnc == 1) {
            b1xm = bX(i, j, k, 0); b1xp = bX(i + 1, j, k, 0); b1ym = bY(i, j, k, 0); b1yp = bY(i, j + 1, k, 0); b1zm = bZ(i, j, k, 0); b1zp = bZ(i, j, k + 1, 0);
        }
        const double aa = has_a ? alpha * A(i, j, k, 0) : 0.0;
        for (int n = 0; n < ncomp; ++n) {
            const int nq = bc.nbc == 1 ? 0 : (n < 3 ? n : 0);
            double cf1 = (j == bc.dlo[1]) ? bc.cflo[nq][1] : 0.0, cf4 = (j == bc.dhi[1]) ? bc.cfhi[nq][1] : 0.0;
            double cf0 = (i == bc.dlo[0]) ? bc.cflo[nq][0] : 0.0, cf3 = (i == bc.dhi[0]) ? bc.cfhi[nq][0] : 0.0;
            double cf2 = (k == bc.dlo[2]) ? bc.cflo[nq][2] : 0.0, cf5 = (k == bc.dhi[2]) ? bc.cfhi[nq][2] : 0.0;
            if (cf) {
                if (i == b.lo[0] && cfm(i - 1, j, k) == 1.0) cf0 = c1x;
                if (i == b.hi[0] && cfm(i + 1, j, k) == 1.0) cf3 = c1x;
                if (j == b.lo[1] && cfm(i, j - 1, k) == 1.0) cf1 = c1y;
                if (j == b.hi[1] && cfm(i, j + 1, k) == 1.0) cf4 = c1y;
                if (k == b.lo[2] && cfm(i, j, k - 1) == 1.0) cf2 = c1z;
                if (k == b.hi[2] && cfm(i, j, k + 1) == 1.0) cf5 = c1z;
            }
            // tens: b holds eta (1 comp) and the 4/3 of the normal component is applied here (x 1.0 otherwise: exact)
            const double sx = (tens && n == 0) ? 4.0 / 3.0 : 1.0, sy = (tens && n == 1) ? 4.0 / 3.0 : 1.0, sz = (tens && n == 2) ? 4.0 / 3.0 : 1.0;
            const bool one = UNI || bnc == 1;
            const double bxm = (one ? b1xm : bX(i, j, k, n)) * sx, bxp = (one ? b1xp : bX(i + 1, j, k, n)) * sx;
            const double bym = (one ? b1ym : bY(i, j, k, n)) * sy, byp = (one ? b1yp : bY(i, j + 1, k, n)) * sy;
            const double bzm = (one ? b1zm : bZ(i, j, k, n)) * sz, bzp = (one ? b1zp : bZ(i, j, k + 1, n)) * sz;
            const double gamma = aa + dhx * (bxm + bxp) + dhy * (bym + byp) + dhz * (bzm + bzp);
            const double g_m_d = gamma - (dhx * (bxm * cf0 + bxp * cf3) + dhy * (bym * cf1 + byp * cf4) + dhz * (bzm * cf2 + bzp * cf5));
            const double p0 = phi(i, j, k, n);
            double pxm = phi(im, j, k, n), pxp = phi(ip, j, k, n), pym = phi(i, jm, k, n), pyp = phi(i, jp, k, n), pzm = phi(i, j, km, n), pzp = phi(i, j, kp, n);
            if (wkp) {
                const WallK& wk = *wkp;
                if (!wk.per[0]) { if (i == bc.dlo[0]) pxm = wallk_ghost(p0, pxp, wk.c1lo[nq][0], wk.c2lo[nq][0]); if (i == bc.dhi[0]) pxp = wallk_ghost(p0, pxm, wk.c1hi[nq][0], wk.c2hi[nq][0]); }
                if (!wk.per[1]) { if (j == bc.dlo[1]) pym = wallk_ghost(p0, pyp, wk.c1lo[nq][1], wk.c2lo[nq][1]); if (j == bc.dhi[1]) pyp = wallk_ghost(p0, pym, wk.c1hi[nq][1], wk.c2hi[nq][1]); }
                if (!wk.per[2]) { if (k == bc.dlo[2]) pzm = wallk_ghost(p0, pzp, wk.c1lo[nq][2], wk.c2lo[nq][2]); if (k == bc.dhi[2]) pzp = wallk_ghost(p0, pzm, wk.c1hi[nq][2], wk.c2hi[nq][2]); }
            }
            const double rho = dhx * (bxm * pxm + bxp * pxp)
                             + dhy * (bym * pym + byp * pyp)
                             + dhz * (bzm * pzm + bzp * pzp);
            const double res = rhs(i, j, k, n) - (gamma * p0 - rho);
            phi(i, j, k, n) = p0 + omega / g_m_d * res;
        }
    }
        return;
    }
    for (int n = 0; n < ncomp; ++n) {
        const int nb = bnc == 1 ? 0 : n;
        const int nq = bc.nbc == 1 ? 0 : (n < 3 ? n : 0);
        const double cf1d = (j == bc.dlo[1]) ? bc.cflo[nq][1] : 0.0, cf4d = (j == bc.dhi[1]) ? bc.cfhi[nq][1] : 0.0;
        for (int k = k0; k <= k1; ++k) {
            const int i = b.lo[0] + 2 * (ih - b.lo[0]) + ((b.lo[0] + j + k + redblack) & 1);
            if (i > b.hi[0]) continue;
            if (shell_only && i > b.lo[0] && i < b.hi[0] && j > b.lo[1] && j < b.hi[1] && k > b.lo[2] && k < b.hi[2]) continue;
            double cf0 = (i == bc.dlo[0]) ? bc.cflo[nq][0] : 0.0, cf3 = (i == bc.dhi[0]) ? bc.cfhi[nq][0] : 0.0;
            double cf2 = (k == bc.dlo[2]) ? bc.cflo[nq][2] : 0.0, cf5 = (k == bc.dhi[2]) ? bc.cfhi[nq][2] : 0.0;
            double cf1 = cf1d, cf4 = cf4d;
            if (cf) {
                if (i == b.lo[0] && cfm(i - 1, j, k) == 1.0) cf0 = c1x;
                if (i == b.hi[0] && cfm(i + 1, j, k) == 1.0) cf3 = c1x;
                if (j == b.lo[1] && cfm(i, j - 1, k) == 1.0) cf1 = c1y;
                if (j == b.hi[1] && cfm(i, j + 1, k) == 1.0) cf4 = c1y;
                if (k == b.lo[2] && cfm(i, j, k - 1) == 1.0) cf2 = c1z;
                if (k == b.hi[2] && cfm(i, j, k + 1) == 1.0) cf5 = c1z;
            }
            const int im = (wrap && i == b.lo[0]) ? b.hi[0] : i - 1, ip = (wrap && i == b.hi[0]) ? b.lo[0] : i + 1;
            const int km = (wrap && k == b.lo[2]) ? b.hi[2] : k - 1, kp = (wrap && k == b.hi[2]) ? b.lo[2] : k + 1;
            // tens: b holds eta (1 comp) and the 4/3 of the normal component is applied here (x 1.0 otherwise: exact)
            const double sx = (tens && n == 0) ? 4.0 / 3.0 : 1.0, sy = (tens && n == 1) ? 4.0 / 3.0 : 1.0, sz = (tens && n == 2) ? 4.0 / 3.0 : 1.0;
            double bxm, bxp, bym, byp, bzm, bzp;
            if (SIG) {                         // mac_bcoef's expression, lower cell first
                const double s0 = bX(i, j, k, sig_comp);
                bxm = sig_scale / (0.5 * (bX(i - 1, j, k, sig_comp) + s0)); bxp = sig_scale / (0.5 * (s0 + bX(i + 1, j, k, sig_comp)));
                bym = sig_scale / (0.5 * (bX(i, j - 1, k, sig_comp) + s0)); byp = sig_scale / (0.5 * (s0 + bX(i, j + 1, k, sig_comp)));
                bzm = sig_scale / (0.5 * (bX(i, j, k - 1, sig_comp) + s0)); bzp = sig_scale / (0.5 * (s0 + bX(i, j, k + 1, sig_comp)));
            } else if (UNI) {
                bxm = bxp = bu.v[0] * sx; bym = byp = bu.v[1] * sy; bzm = bzp = bu.v[2] * sz;
            } else {
                bxm = bX(i, j, k, nb) * sx; bxp = bX(i + 1, j, k, nb) * sx;
                bym = bY(i, j, k, nb) * sy; byp = bY(i, j + 1, k, nb) * sy;
                bzm = bZ(i, j, k, nb) * sz; bzp = bZ(i, j, k + 1, nb) * sz;
            }
            const double aa = has_a ? alpha * A(i, j, k, 0) : 0.0;
            const double gamma = aa + dhx * (bxm + bxp) + dhy * (bym + byp) + dhz * (bzm + bzp);
            const double g_m_d = gamma - (dhx * (bxm * cf0 + bxp * cf3) + dhy * (bym * cf1 + byp * cf4) + dhz * (bzm * cf2 + bzp * cf5));
            const double p0 = phi(i, j, k, n);
            double pxm = phi(im, j, k, n), pxp = phi(ip, j, k, n), pym = phi(i, jm, k, n), pyp = phi(i, jp, k, n), pzm = phi(i, j, km, n), pzp = phi(i, j, kp, n);
            if (wkp) {
                const WallK& wk = *wkp;
                if (!wk.per[0]) { if (i == bc.dlo[0]) pxm = wallk_ghost(p0, pxp, wk.c1lo[nq][0], wk.c2lo[nq][0]); if (i == bc.dhi[0]) pxp = wallk_ghost(p0, pxm, wk.c1hi[nq][0], wk.c2hi[nq][0]); }
                if (!wk.per[1]) { if (j == bc.dlo[1]) pym = wallk_ghost(p0, pyp, wk.c1lo[nq][1], wk.c2lo[nq][1]); if (j == bc.dhi[1]) pyp = wallk_ghost(p0, pym, wk.c1hi[nq][1], wk.c2hi[nq][1]); }
                if (!wk.per[2]) { if (k == bc.dlo[2]) pzm = wallk_ghost(p0, pzp, wk.c1lo[nq][2], wk.c2lo[nq][2]); if (k == bc.dhi[2]) pzp = wallk_ghost(p0, pzm, wk.c1hi[nq][2], wk.c2hi[nq][2]); }
            }
            const double rho = dhx * (bxm * pxm + bxp * pxp)
                             + dhy * (bym * pym + byp * pyp)
                             + dhz * (bzm * pzm + bzp * pzp);
            const double res = rhs(i, j, k, n) - (gamma * p0 - rho);
            const double pn = p0 + omega / g_m_d * res;
            phi(i, j, k, n) = pn;
            if (cf && cfc.maintain && (i <= b.lo[0] + 1 || i >= b.hi[0] - 1 || j <= b.lo[1] + 1 || j >= b.hi[1] - 1 || k <= b.lo[2] + 1 || k >= b.hi[2] - 1))
                cf_maintain(phi, cfm, b, cfc, i, j, k, pn);
        }
    }
}

// ---------------------------------------------------------------------------- GSRB, one component, software-pipelined
// The colour pass of the scalar solves (MAC projection, scalar diffusion) on levels without coarse/fine faces.  In k_abec_gsrb the
// compiler cannot move the loads of plane k + 1 above the store of plane k (same array), so every thread pays one full memory
// round trip per plane.  A colour pass has no such hazard -- the neighbours of a cell of the active colour have the other colour and
// the cell itself is written by its own thread only -- so this kernel loads the operands of NP planes, then updates them: NP round
// trips in flight per wavefront.  Same expressions as k_abec_gsrb (same doubles).
template <int BMODE>
struct Gs1Ops {
    double pc, pxm, pxp, pym, pyp, pzm, pzp, r, a;
    double c[BMODE == 2 ? 1 : (BMODE == 1 ? 7 : 6)];   // BMODE 0: the six face coefficients; 1: sigma at the cell and its six neighbours
    int i;
};
template <int BMODE, int NP, bool CF>
__global__ void __launch_bounds__(256) k_abec_gsrb1(Tiling t, const BoxD* __restrict__ boxes,
    const FabD* __restrict__ phit, const FabD* __restrict__ rhst, const FabD* __restrict__ at,
    const FabD* __restrict__ bxt, const FabD* __restrict__ byt, const FabD* __restrict__ bzt,
    double alpha, double dhx, double dhy, double dhz, int redblack, double omega, GsrbBC bc, int wrap, int sig_comp, double sig_scale, BUni bu,
    const FabD* __restrict__ cfmt, CfC1 cfc, int zero, const WallK* __restrict__ wkp = nullptr)
{
    const int fab = tile_fab(t);
    const BoxD b = boxes[fab];
    BoxD hb = b;
    hb.hi[0] = b.lo[0] + (b.len(0) + 1) / 2 - 1;
    int ih, j, k0, k1;
    if (!tile_ijk(t, hb, ih, j, k0, k1)) return;
    const FabD phi = phit[fab], rhs = rhst[fab], bX = bxt[fab], bY = byt[fab], bZ = bzt[fab];
    const bool has_a = (at != nullptr) && alpha != 0.0;
    FabD A; if (has_a) A = at[fab];
    // coarse/fine faces as in k_abec_gsrb (a ghost cell written by cf_maintain is not read in the same pass, see the top of the file)
    constexpr bool cf = CF;                 // compile-time: the instantiation for levels without coarse/fine faces carries none of this
    FabD cfm; if (cf) cfm = cfmt[fab];
    CfDir cd;
    if (cf) cd = cf_dir(b, cfc);
    const double c1x = cf ? cd.c1[0] : 0.0, c1y = cf ? cd.c1[1] : 0.0, c1z = cf ? cd.c1[2] : 0.0;
    const int jm = (wrap && j == b.lo[1]) ? b.hi[1] : j - 1, jp = (wrap && j == b.hi[1]) ? b.lo[1] : j + 1;
    const double cf1 = (j == bc.dlo[1]) ? bc.cflo[0][1] : 0.0, cf4 = (j == bc.dhi[1]) ? bc.cfhi[0][1] : 0.0;
    auto load = [&](int k, Gs1Ops<BMODE>& o) {
        const int i = b.lo[0] + 2 * (ih - b.lo[0]) + ((b.lo[0] + j + k + redblack) & 1);
        o.i = i;
        if (i > b.hi[0]) return;
        const int im = (wrap && i == b.lo[0]) ? b.hi[0] : i - 1, ip = (wrap && i == b.hi[0]) ? b.lo[0] : i + 1;
        const int km = (wrap && k == b.lo[2]) ? b.hi[2] : k - 1, kp = (wrap && k == b.hi[2]) ? b.lo[2] : k + 1;
        if (zero) o.pc = o.pxm = o.pxp = o.pym = o.pyp = o.pzm = o.pzp = 0.0;      // phi_is_zero: nothing to read
        else {
            o.pc = phi(i, j, k, 0);
            o.pxm = phi(im, j, k, 0); o.pxp = phi(ip, j, k, 0);
            o.pym = phi(i, jm, k, 0); o.pyp = phi(i, jp, k, 0);
            o.pzm = phi(i, j, km, 0); o.pzp = phi(i, j, kp, 0);
            if (wkp) {            // walls inside the kernel (WallK): the ghost formula on the values just read
                const WallK& wk = *wkp;
                if (!wk.per[0]) { if (i == bc.dlo[0]) o.pxm = wallk_ghost(o.pc, o.pxp, wk.c1lo[0][0], wk.c2lo[0][0]); if (i == bc.dhi[0]) o.pxp = wallk_ghost(o.pc, o.pxm, wk.c1hi[0][0], wk.c2hi[0][0]); }
                if (!wk.per[1]) { if (j == bc.dlo[1]) o.pym = wallk_ghost(o.pc, o.pyp, wk.c1lo[0][1], wk.c2lo[0][1]); if (j == bc.dhi[1]) o.pyp = wallk_ghost(o.pc, o.pym, wk.c1hi[0][1], wk.c2hi[0][1]); }
                if (!wk.per[2]) { if (k == bc.dlo[2]) o.pzm = wallk_ghost(o.pc, o.pzp, wk.c1lo[0][2], wk.c2lo[0][2]); if (k == bc.dhi[2]) o.pzp = wallk_ghost(o.pc, o.pzm, wk.c1hi[0][2], wk.c2hi[0][2]); }
            }
        }
        o.r = rhs(i, j, k, 0);
        o.a = has_a ? A(i, j, k, 0) : 0.0;
        if (BMODE == 0) {
            o.c[0] = bX(i, j, k, 0); o.c[1] = bX(i + 1, j, k, 0); o.c[2] = bY(i, j, k, 0); o.c[3] = bY(i, j + 1, k, 0);
            o.c[4] = bZ(i, j, k, 0); o.c[5] = bZ(i, j, k + 1, 0);
        } else if (BMODE == 1) {
            o.c[0] = bX(i, j, k, sig_comp);
            o.c[1] = bX(i - 1, j, k, sig_comp); o.c[2] = bX(i + 1, j, k, sig_comp);
            o.c[3] = bX(i, j - 1, k, sig_comp); o.c[4] = bX(i, j + 1, k, sig_comp);
            o.c[5] = bX(i, j, k - 1, sig_comp); o.c[6] = bX(i, j, k + 1, sig_comp);
        }
    };
    auto update = [&](int k, const Gs1Ops<BMODE>& o) {
        const int i = o.i;
        if (zero) {                          // the other cell of the thread's pair (lo + 2m, lo + 2m + 1): zero
            const int iL = b.lo[0] + 2 * (ih - b.lo[0]), io = i == iL ? iL + 1 : iL;
            if (io <= b.hi[0]) phi(io, j, k, 0) = 0.0;
        }
        if (i > b.hi[0]) return;
        double bxm, bxp, bym, byp, bzm, bzp;
        if (BMODE == 0) { bxm = o.c[0]; bxp = o.c[1]; bym = o.c[2]; byp = o.c[3]; bzm = o.c[4]; bzp = o.c[5]; }
        else if (BMODE == 1) {                 // mac_bcoef's expression, lower cell first
            const double s0 = o.c[0];
            bxm = sig_scale / (0.5 * (o.c[1] + s0)); bxp = sig_scale / (0.5 * (s0 + o.c[2]));
            bym = sig_scale / (0.5 * (o.c[3] + s0)); byp = sig_scale / (0.5 * (s0 + o.c[4]));
            bzm = sig_scale / (0.5 * (o.c[5] + s0)); bzp = sig_scale / (0.5 * (s0 + o.c[6]));
        } else { bxm = bxp = bu.v[0]; bym = byp = bu.v[1]; bzm = bzp = bu.v[2]; }
        double cf0 = (i == bc.dlo[0]) ? bc.cflo[0][0] : 0.0, cf3 = (i == bc.dhi[0]) ? bc.cfhi[0][0] : 0.0;
        double cf2 = (k == bc.dlo[2]) ? bc.cflo[0][2] : 0.0, cf5 = (k == bc.dhi[2]) ? bc.cfhi[0][2] : 0.0;
        double c1 = cf1, c4 = cf4;
        if (cf) {
            if (i == b.lo[0] && cfm(i - 1, j, k) == 1.0) cf0 = c1x;
            if (i == b.hi[0] && cfm(i + 1, j, k) == 1.0) cf3 = c1x;
            if (j == b.lo[1] && cfm(i, j - 1, k) == 1.0) c1 = c1y;
            if (j == b.hi[1] && cfm(i, j + 1, k) == 1.0) c4 = c1y;
            if (k == b.lo[2] && cfm(i, j, k - 1) == 1.0) cf2 = c1z;
            if (k == b.hi[2] && cfm(i, j, k + 1) == 1.0) cf5 = c1z;
        }
        const double aa = has_a ? alpha * o.a : 0.0;
        const double gamma = aa + dhx * (bxm + bxp) + dhy * (bym + byp) + dhz * (bzm + bzp);
        const double g_m_d = gamma - (dhx * (bxm * cf0 + bxp * cf3) + dhy * (bym * c1 + byp * c4) + dhz * (bzm * cf2 + bzp * cf5));
        const double rho = dhx * (bxm * o.pxm + bxp * o.pxp) + dhy * (bym * o.pym + byp * o.pyp) + dhz * (bzm * o.pzm + bzp * o.pzp);
        const double res = o.r - (gamma * o.pc - rho);
        const double pn = o.pc + omega / g_m_d * res;
        phi(i, j, k, 0) = pn;
        if (cf && cfc.maintain && (i <= b.lo[0] + 1 || i >= b.hi[0] - 1 || j <= b.lo[1] + 1 || j >= b.hi[1] - 1 || k <= b.lo[2] + 1 || k >= b.hi[2] - 1)) {
            const double nb[6] = {o.pxm, o.pxp, o.pym, o.pyp, o.pzm, o.pzp};
            cf_maintain_nb(phi, cfm, b, cd, i, j, k, pn, nb);
        }
    };
    for (int k = k0; k <= k1; k += NP) {
        Gs1Ops<BMODE> ops[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) if (k + q <= k1) load(k + q, ops[q]);
#pragma unroll
        for (int q = 0; q < NP; ++q) if (k + q <= k1) update(k + q, ops[q]);
    }
}

// ---------------------------------------------------------------------------- GSRB, one component, pair-marching
// A thread owns the two cells (lo + 2m, lo + 2m + 1) of a row and marches in z.  In every plane one of the two has the active colour.
// The row pairs of phi (and of sigma, BMODE 1) are read with one 16-byte load per plane -- every byte of every cache line it touches is
// used -- and kept in registers for three planes, so the z-neighbours cost no loads and the x-neighbours come from the pair itself or
// from the adjacent lane; only the y-neighbours and the right-hand side are 8-byte loads at the active cell.  7 load instructions per
// updated cell instead of 15, less than half the L1 / L2 traffic (the colour passes of k_abec_gsrb1 ran at L2 rather than HBM speed:
// every row was fetched for three planes and by three rows of threads).  No hazards: a colour pass only writes cells of its colour and
// only reads, besides the cell itself, cells of the other one.  Same expressions as k_abec_gsrb (same doubles).
// BMODE 1 (AbecCoef::sig) or 2 (AbecCoef::b_uniform).
struct D2 { double l, r; };
__device__ __forceinline__ D2 ld2(const FabD& f, int i, int j, int k, int n)
{
    typedef double v2u __attribute__((ext_vector_type(2), aligned(8)));
    const v2u v = *(const __attribute__((address_space(1))) v2u*)(f.gp() + f.off(i, j, k) + f.cs * n);
    D2 r; r.l = v.x; r.r = v.y;
    return r;
}
// coarse/fine variants: at least 4 wavefronts per SIMD (<= 128 VGPRs; they need 132 / 117 unconstrained and ran at 3: 204 us per 256^3 pass
// against 124 us for the variant without coarse/fine faces at 5)
// ALLCF: every ghost cell beyond a face of the box is a coarse/fine ghost cell (a refined level that is one box strictly inside the domain):
// the masks are the constant 1 and are not loaded
template <int BMODE, bool CF, bool MAINT, bool ALLCF = false>
__global__ void __launch_bounds__(256, (CF ? 4 : 1)) k_abec_gsrb2(Tiling t, const BoxD* __restrict__ boxes,
    const FabD* __restrict__ phit, const FabD* __restrict__ rhst, const FabD* __restrict__ at, const FabD* __restrict__ sgt,
    double alpha, double dhx, double dhy, double dhz, int redblack, double omega, GsrbBC bc, int wrap, int sig_comp, double sig_scale, BUni bu,
    const FabD* __restrict__ cfmt, CfC1 cfc, int zero, int comp, int bq)
{
    const int fab = tile_fab(t);
    const BoxD b = boxes[fab];
    BoxD hb = b;
    hb.hi[0] = b.lo[0] + (b.len(0) + 1) / 2 - 1;
    int ih, j, k0, k1;
    if (!tile_ijk(t, hb, ih, j, k0, k1)) return;
    const FabD phi = phit[fab], rhs = rhst[fab];
    // coarse/fine faces as in k_abec_gsrb: the ghost formula's first-interior-cell weight, and the ghost cells kept current (cf_maintain)
    constexpr bool cf = CF;
    FabD cfm; if (cf) cfm = cfmt[fab];
    CfDir cd;
    if (cf) cd = cf_dir(b, cfc);
    const double c1x = cf ? cd.c1[0] : 0.0, c1y = cf ? cd.c1[1] : 0.0, c1z = cf ? cd.c1[2] : 0.0;
    FabD S; if (BMODE == 1) S = sgt[fab];
    const bool has_a = (at != nullptr) && alpha != 0.0;
    FabD A; if (has_a) A = at[fab];
    const int bx = 1 << tile_bxs(t), tx = (int)threadIdx.x & (bx - 1);
    const int iL = b.lo[0] + 2 * (ih - b.lo[0]), iR = iL + 1;
    // the neighbour lane of the row exists (same wavefront, same row, inside the box)?
    const bool laneL = tx > 0 && iL > b.lo[0], laneR = tx < bx - 1 && tx < 63 && iR + 1 <= b.hi[0];
    const int jm = (wrap && j == b.lo[1]) ? b.hi[1] : j - 1, jp = (wrap && j == b.hi[1]) ? b.lo[1] : j + 1;
    const double cf1 = (j == bc.dlo[1]) ? bc.cflo[bq][1] : 0.0, cf4 = (j == bc.dhi[1]) ? bc.cfhi[bq][1] : 0.0;
    auto kw = [&](int k) { return wrap ? (k < b.lo[2] ? b.hi[2] : (k > b.hi[2] ? b.lo[2] : k)) : k; };
    // zero (phi_is_zero, wrap only): phi is not read; every cell of the pair is written
    D2 pb, pc;
    if (zero) { pb.l = pb.r = pc.l = pc.r = 0.0; }
    else { pb = ld2(phi, iL, j, kw(k0 - 1), comp); pc = ld2(phi, iL, j, k0, comp); }
    D2 sb, sc;
    if (BMODE == 1) { sb = ld2(S, iL, j, k0 - 1, sig_comp); sc = ld2(S, iL, j, k0, sig_comp); }
    // y-neighbours and right-hand side of the active cell: loaded one plane ahead like the row pair (the pass is bound by the memory latency
    // of what a thread has in flight, not by bytes: with these three loads issued behind the arithmetic of every plane a constant-coefficient
    // pass took as long as the MAC form that reads twice as much)
    auto yload = [&](int k, double& ym, double& yp, double& rv) {
        const int i = iL + ((b.lo[0] + j + k + redblack) & 1);
        if (i <= b.hi[0] && k <= k1) {
            ym = zero ? 0.0 : (double)phi(i, jm, k, comp); yp = zero ? 0.0 : (double)phi(i, jp, k, comp);
            rv = rhs(i, j, k, comp);
        } else { ym = yp = rv = 0.0; }
    };
    double nym, nyp, nrr;
    yload(k0, nym, nyp, nrr);
    for (int k = k0; k <= k1; ++k) {
        const int par = (b.lo[0] + j + k + redblack) & 1;         // 0: the left cell of the pair is active
        const int i = iL + par;
        const bool live = i <= b.hi[0];
        const double pym = nym, pyp = nyp, rr = nrr;
        yload(k + 1, nym, nyp, nrr);
        D2 pa;
        if (zero) { pa.l = pa.r = 0.0; } else pa = ld2(phi, iL, j, kw(k + 1), comp);
        D2 sa;
        if (BMODE == 1) sa = ld2(S, iL, j, k + 1, sig_comp);
        // x-neighbours: the other cell of the pair, and the adjacent lane's near cell (or a load where there is no such lane)
        const double fromL = __shfl_up(pc.r, 1, 64), fromR = __shfl_down(pc.l, 1, 64);
        double pxm, pxp;
        if (zero) { pxm = pxp = 0.0; }
        else if (par == 0) {
            pxp = (wrap && i == b.hi[0]) ? (double)phi(b.lo[0], j, k, comp) : pc.r;
            pxm = laneL ? fromL : (double)phi((wrap && i == b.lo[0]) ? b.hi[0] : i - 1, j, k, comp);
        } else {
            pxm = pc.l;
            pxp = laneR ? fromR : (live ? (double)phi((wrap && i == b.hi[0]) ? b.lo[0] : i + 1, j, k, comp) : 0.0);
        }
        // coarse/fine masks of the (up to six) ghost cells this cell's coefficient or its ghost rewrite can need: loaded here, with the data
        // of the plane, not behind the arithmetic that uses them (they used to be two dependent round trips per plane for every wavefront)
        double mxl = 0.0, mxh = 0.0, myl = 0.0, myh = 0.0, mzl = 0.0, mzh = 0.0;
        if (cf && live) {
            if constexpr (ALLCF) {
                mxl = i - b.lo[0] <= 1 ? 1.0 : 0.0; mxh = b.hi[0] - i <= 1 ? 1.0 : 0.0;
                myl = j - b.lo[1] <= 1 ? 1.0 : 0.0; myh = b.hi[1] - j <= 1 ? 1.0 : 0.0;
                mzl = k - b.lo[2] <= 1 ? 1.0 : 0.0; mzh = b.hi[2] - k <= 1 ? 1.0 : 0.0;
            } else {
            if (i - b.lo[0] <= 1) mxl = cfm(b.lo[0] - 1, j, k);
            if (b.hi[0] - i <= 1) mxh = cfm(b.hi[0] + 1, j, k);
            if (j - b.lo[1] <= 1) myl = cfm(i, b.lo[1] - 1, k);
            if (b.hi[1] - j <= 1) myh = cfm(i, b.hi[1] + 1, k);
            if (k - b.lo[2] <= 1) mzl = cfm(i, j, b.lo[2] - 1);
            if (b.hi[2] - k <= 1) mzh = cfm(i, j, b.hi[2] + 1);
            }
        }
        double sxm = 0.0, sxp = 0.0, sym = 0.0, syp = 0.0;
        if (BMODE == 1) {
            const double sfl = __shfl_up(sc.r, 1, 64), sfr = __shfl_down(sc.l, 1, 64);
            if (par == 0) { sxp = sc.r; sxm = laneL ? sfl : (double)S(i - 1, j, k, sig_comp); }
            else { sxm = sc.l; sxp = laneR ? sfr : (live ? (double)S(i + 1, j, k, sig_comp) : 0.0); }
        }
        if (live) {
            const double p0 = par ? pc.r : pc.l;
            const double pzm = par ? pb.r : pb.l, pzp = par ? pa.r : pa.l;
            double bxm, bxp, bym, byp, bzm, bzp;
            if (BMODE == 1) {                  // mac_bcoef's expression, lower cell first
                sym = S(i, j - 1, k, sig_comp); syp = S(i, j + 1, k, sig_comp);
                const double s0 = par ? sc.r : sc.l, szm = par ? sb.r : sb.l, szp = par ? sa.r : sa.l;
                bxm = sig_scale / (0.5 * (sxm + s0)); bxp = sig_scale / (0.5 * (s0 + sxp));
                bym = sig_scale / (0.5 * (sym + s0)); byp = sig_scale / (0.5 * (s0 + syp));
                bzm = sig_scale / (0.5 * (szm + s0)); bzp = sig_scale / (0.5 * (s0 + szp));
            } else { bxm = bxp = bu.v[0]; bym = byp = bu.v[1]; bzm = bzp = bu.v[2]; }
            double cf0 = (i == bc.dlo[0]) ? bc.cflo[bq][0] : 0.0, cf3 = (i == bc.dhi[0]) ? bc.cfhi[bq][0] : 0.0;
            double cf2 = (k == bc.dlo[2]) ? bc.cflo[bq][2] : 0.0, cf5 = (k == bc.dhi[2]) ? bc.cfhi[bq][2] : 0.0;
            double c1 = cf1, c4 = cf4;
            if (cf) {
                if (i == b.lo[0] && mxl == 1.0) cf0 = c1x;
                if (i == b.hi[0] && mxh == 1.0) cf3 = c1x;
                if (j == b.lo[1] && myl == 1.0) c1 = c1y;
                if (j == b.hi[1] && myh == 1.0) c4 = c1y;
                if (k == b.lo[2] && mzl == 1.0) cf2 = c1z;
                if (k == b.hi[2] && mzh == 1.0) cf5 = c1z;
            }
            const double aa = has_a ? alpha * A(i, j, k, 0) : 0.0;
            const double gamma = aa + dhx * (bxm + bxp) + dhy * (bym + byp) + dhz * (bzm + bzp);
            const double g_m_d = gamma - (dhx * (bxm * cf0 + bxp * cf3) + dhy * (bym * c1 + byp * c4) + dhz * (bzm * cf2 + bzp * cf5));
            const double rho = dhx * (bxm * pxm + bxp * pxp) + dhy * (bym * pym + byp * pyp) + dhz * (bzm * pzm + bzp * pzp);
            const double res = rr - (gamma * p0 - rho);
            const double pn = p0 + omega / g_m_d * res;
            phi(i, j, k, comp) = pn;
            if (cf && MAINT) {
                // cf_maintain_nb with the masks already in registers: (distance from the face, mask, neighbour towards / away from the face)
                auto rewrite = [&](int d, int dist, double mask, double towards, double away, int gi, int gj, int gk) {
                    const int NX = cd.nx[d];
                    if (NX < 2 || mask != 1.0 || dist + 1 != (NX == 2 ? 1 : 2)) return;
                    double v = 0.0;
                    if (NX == 2) v += pn * cd.c1[d];
                    else { v += towards * cd.c1[d]; v += pn * cd.c2[d]; if (NX > 3) v += away * cd.c3[d]; }
                    phi(gi, gj, gk, comp) = v;
                };
                rewrite(0, i - b.lo[0], mxl, pxm, pxp, b.lo[0] - 1, j, k);
                rewrite(0, b.hi[0] - i, mxh, pxp, pxm, b.hi[0] + 1, j, k);
                rewrite(1, j - b.lo[1], myl, pym, pyp, i, b.lo[1] - 1, k);
                rewrite(1, b.hi[1] - j, myh, pyp, pym, i, b.hi[1] + 1, k);
                rewrite(2, k - b.lo[2], mzl, pzm, pzp, i, j, b.lo[2] - 1);
                rewrite(2, b.hi[2] - k, mzh, pzp, pzm, i, j, b.hi[2] + 1);
            }
        }
        if (zero) { const int io = par ? iL : iR; if (io <= b.hi[0]) phi(io, j, k, comp) = 0.0; }
        pb = pc; pc = pa;
        if (BMODE == 1) { sb = sc; sc = sa; }
    }
}

// ---------------------------------------------------------------------------- one-pass red + black sweep, index-wrap levels
// k_abec_gsrb_rb: a red AND a black colour pass of a level that is one box spanning a periodic domain in ONE launch, out of place
// (pin -> pout), with the doubles of the two k_abec_gsrb2 launches.  Why: on the natural layout a colour pass reads whole phi lines,
// writes half of every phi line (a read-modify-write at the memory) and reads half-used rhs lines -- 536 MB (+ 134 MB of density in the
// MAC form) per pass at 256^3, whatever the kernel does.  The sweep reads phi, rhs (and the density) once and writes every phi line
// whole: 402 (536) MB per SWEEP.
// A workgroup of NW wavefronts owns RW = NW / wpr rows of the box (wpr wavefronts per row: 128 cells each), rows 0 and RW - 1 as halo,
// and marches through tz planes.  A thread owns the cell pair (lo + 2 ih, lo + 2 ih + 1) of its row and holds the pin / density / rhs
// pairs of the planes around the current one in registers; these 16-byte loads, a plane ahead, are the ONLY global loads of the loop (the
// halo rows add the 8-byte loads of the row outside the tile).  Everything a cell needs from another thread travels through lane shifts
// (the pair next to it) and LDS: per plane p every thread publishes the old value and density of its BLACK cell of plane p + 1 (the x- and
// y-neighbours of the red cells of that plane), forms the red update of plane p, publishes the new red value and the density there, and --
// one barrier later -- forms the black update of plane p - 1 (rows 1 .. RW - 2) from the new reds around it: its pair's, the planes p - 2
// and p of its own column (registers), the next pair's (lane shift; across the 128-cell boundary: LDS), the rows above and below (LDS).
// (A load that is issued and consumed inside one iteration waits for every earlier load of the in-order queue, the prefetches included:
// the first version, with such loads for the lane-0 / lane-63 neighbours and the black cell's densities, took 226 us per sweep.)
// Every workgroup recomputes the reds of the plane below and above its chunk and of its two halo rows.
__device__ __forceinline__ double rb_lane_lo(double v)      // the value lane - 1 holds (DPP wave_shr:1)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rb_lane_hi(double v)      // the value lane + 1 holds (DPP wave_shl:1)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rb_take(double v) { double r; asm volatile("v_mov_b64 %0, %1" : "=v"(r) : "v"(v)); return r; }

// WALLS: the box spans a domain with non-periodic directions (no coarse/fine faces).  Nothing is read from phi's ghost cells: the value
// beyond a domain face is the ghost formula of k_abec_bc on the values at hand, g = p0 * c1 + p_in * c2 (Neumann 1, 0; reflect-odd -1, 0;
// homogeneous Dirichlet of order <= 3: the two Lagrange weights; p_in: the neighbour on the inner side -- for a black cell the NEW red,
// as after the ghost fill between the colours), gamma loses dh * b * c1 there, the density beyond the face comes from its (filled) ghost
// cell, and a black cell at an x-face forms that face's coefficient itself.  Rows / planes outside the box that a tile carries as halo
// compute on whatever the ghost cells hold; every use of their values is replaced as above.
// NBR ("neighbours"): the boxes of a level that covers its domain in several boxes (one launch: blockIdx.y = local box; a rank that owns
// one box of a sharded level).  A face of a box is either a domain wall (the formula above) or OPEN: another box or a periodic image lies
// behind it.  phi comes with TWO filled ghost layers (one FillBoundary per sweep instead of one per colour) and the small launch
// k_abec_rb_ghost in front of this kernel has replaced every RED ghost cell next to an open face by its updated value (in place: nobody
// else reads the old value of a red cell) -- the black update of a cell at an open face needs exactly that value.  Here the rows / planes a
// tile carries outside its box therefore pass their ghost values through instead of updating them, and the first / last lane of a row loads
// the ghost column beyond an open x-face a plane ahead: its old black value for the red update, its new red value for the black update.
// rhs (and the a-term) come with one filled ghost layer, the density with two (k_abec_rb_ghost forms the face coefficients of the ghost cell).
struct RbBC { int per[3]; double c1lo[3], c2lo[3], c1hi[3], c2hi[3]; int dlo[3], dhi[3]; double c3lo[3] = {0.0, 0.0, 0.0}, c3hi[3] = {0.0, 0.0, 0.0}; };
#ifndef IAMRX_RBW_EXP
#define IAMRX_RBW_EXP 0      // timing experiments on the wall variant (wrong results): 1: no density loads beyond x-faces, 2: no wall terms in gamma, 4: no y / z wall ghosts, 8: no x-wall branch, 16: no face coefficient in the black x-wall branch, 32: no x wall weights, 64: no wall density in the red x-wall branch
#endif

// one Gauss-Seidel update of the sweep kernels: k_abec_gsrb2's expressions (no coarse-fine terms)
template <bool SG, bool WALLS>
__device__ __forceinline__ double rb_update(double dhx, double dhy, double dhz, double omega,
    double p0, double pxm, double pxp, double pym, double pyp, double pzm, double pzp, double rr, double aa,
    double bxm, double bxp, double bym, double byp, double bzm, double bzp,
    double cxl, double cxh, double cyl, double cyh, double czl, double czh)
{
    const double gamma = aa + dhx * (bxm + bxp) + dhy * (bym + byp) + dhz * (bzm + bzp);
    // k_abec_gsrb2 subtracts the wall terms dhx (bxm c_lo + bxp c_hi) + dhy (...) + dhz (...) from gamma here, c = c1 at a domain face the
    // cell touches and 0 elsewhere: the terms with c = 0 are +-0 for finite coefficients and add nothing -- an index-wrap level keeps
    // gamma; a level with walls evaluates the reference's expression itself with the six weights of the cell (zero away from the walls).
    // (Round 4 selected the terms by face flags instead -- `ws += f == 0 ? 0 : dh (b c)` -- which cost the density form 40 us of its 205 us
    // per 256^3 sweep: tools/r5_rbw_exp.sh)
    double g_m_d = gamma;
    if (WALLS && !(IAMRX_RBW_EXP & 2)) g_m_d = gamma - (dhx * (bxm * cxl + bxp * cxh) + dhy * (bym * cyl + byp * cyh) + dhz * (bzm * czl + bzp * czh));
    const double rho = dhx * (bxm * pxm + bxp * pxp) + dhy * (bym * pym + byp * pyp) + dhz * (bzm * pzm + bzp * pzp);
    const double res = rr - (gamma * p0 - rho);
    return p0 + omega / g_m_d * res;
}

// W3 (with WALLS, not NBR): the ghost formula has a THIRD weight, g = p0 c1 + p_in c2 + p_in2 c3 -- the homogeneous coarse/fine ghost value of
// order 4 (k_cf_fill: the face cell, the cell behind it and the one behind that; for a red cell old black and old red, for a black cell the
// new red and the old black: what a fill between the colours would have read).  p_in2 comes from the lane next to the face lane (x), from
// 16-byte loads a plane ahead by the two face rows (y), and from the row registers or two direct loads at the box's last planes (z).
// ACC: the sweep is the LAST one of a V-cycle on the finest level and `pout` is the SOLUTION: the kernel stores pout + (the swept correction)
// instead of the correction -- the `sol += cor` pass behind the V-cycle (24 bytes per cell) becomes 8 more bytes read by this launch; the
// solution's row pairs are fetched a plane ahead like every other load.  Same doubles (sol + 1.0 * cor).
template <int BMODE, int NW, bool HASA, bool WALLS = false, bool NBR = false, bool XO = false, bool W3 = false, bool ACC = false>
__global__ void __launch_bounds__(64 * NW) k_abec_gsrb_rb(BoxD b, FabD pin, FabD pout, FabD rhs, FabD A, FabD S, double alpha,
    double dhx, double dhy, double dhz, double omega, int sig_comp, double sig_scale, BUni bu, int wpr, int tz, int nty, int zero, int comp, int xcd_chunk,
    RbBC bc = RbBC(), const BoxD* __restrict__ boxes = nullptr, const FabD* __restrict__ pint = nullptr, const FabD* __restrict__ poutt = nullptr,
    const FabD* __restrict__ rhst = nullptr, const FabD* __restrict__ At = nullptr, const FabD* __restrict__ St = nullptr, int sel = 0)
{
#if defined(__HIP_DEVICE_COMPILE__)    // (the host pass has no global address space: FabD::gp)
    constexpr bool SG = BMODE == 1;
    if (NBR) {
        const int fab = (int)blockIdx.y;
        b = boxes[fab]; pin = pint[fab]; pout = poutt[fab]; rhs = rhst[fab];
        if (HASA) A = At[fab];
        if (SG) S = St[fab];
    }
    __shared__ double RED[3][NW][64];                        // new red values of the planes q - 2, q - 1, q
    __shared__ double YM[SG ? 3 : 1][SG ? NW : 1][64];       // the face coefficients below / above those cells (every face of the grid lies
    __shared__ double YP[SG ? 3 : 1][SG ? NW : 1][64];       // between a red and a black cell: the black update divides nothing but omega / gamma)
    __shared__ double XF[SG ? 3 : 1][SG ? NW : 1][2];        // lanes 0 / 63: the coefficient of the face towards the neighbouring wavefront's pair
    __shared__ double BPH[2][NW][64];                        // old values of the black cells of the planes q, q + 1
    __shared__ double BSG[SG ? 2 : 1][SG ? NW : 1][64];      // the density at those cells
    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rw = NW / wpr, r = w / wpr, xw = w - r * wpr;
    // workgroup -> tile: consecutive workgroups go to different XCDs (8, each with its own L2); tiles that are neighbours in y (they share
    // their halo rows) are handed to one XCD: XCD x runs the tiles x * xcd_chunk ... of the order "y-tile fastest"
    const int tile = xcd_chunk > 0 ? ((int)blockIdx.x & 7) * xcd_chunk + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int ty = tile % nty, tzc = tile / nty;
    const int ny = b.len(1), nz = b.len(2);
    const int k0 = b.lo[2] + tzc * tz;
    if (k0 > b.hi[2]) return;
    const int kend = min(k0 + tz - 1, b.hi[2]);
    if (NBR && sel != 0) {
        // sel 1: only the tiles that read nothing outside the box (no ghost cell of phi, rhs or the coefficients: they can run while the
        // ghost layers are still being exchanged); sel 2: only the others.  XO: every tile reads the ghost columns.
        const bool bnd = XO || ty == 0 || b.lo[1] + (ty + 1) * (rw - 2) + 1 > b.hi[1] || tzc == 0 || kend + 2 > b.hi[2];
        if ((sel == 1) == bnd) return;
    }
    // walls of this box, per face (NBR: where the box touches a non-periodic side of the domain; else both faces of a non-periodic direction)
    const bool wxl = WALLS && !bc.per[0] && (!NBR || b.lo[0] == bc.dlo[0]), wxh = WALLS && !bc.per[0] && (!NBR || b.hi[0] == bc.dhi[0]);
    const bool wyl = WALLS && !bc.per[1] && (!NBR || b.lo[1] == bc.dlo[1]), wyh = WALLS && !bc.per[1] && (!NBR || b.hi[1] == bc.dhi[1]);
    const bool wzl = WALLS && !bc.per[2] && (!NBR || b.lo[2] == bc.dlo[2]), wzh = WALLS && !bc.per[2] && (!NBR || b.hi[2] == bc.dhi[2]);
    const bool wall1 = NBR || (WALLS && !bc.per[1]), wall2 = NBR || (WALLS && !bc.per[2]);      // no index wrap in y / z: ghost rows / planes
    auto wj = [&](int j) { if (wall1) return min(max(j, b.lo[1] - 1), b.hi[1] + 1); return j < b.lo[1] ? j + ny : (j > b.hi[1] ? j - ny : j); };
    auto wk = [&](int k) { if (wall2) return min(max(k, b.lo[2] - 1), b.hi[2] + 1); return k < b.lo[2] ? k + nz : (k > b.hi[2] ? k - nz : k); };
    const int jraw = b.lo[1] + ty * (rw - 2) + (r - 1);
    const bool row_out = NBR && (jraw < b.lo[1] || jraw > b.hi[1]);      // a ghost row: its reds are k_abec_rb_ghost's
    const bool owner = r >= 1 && r <= rw - 2 && jraw <= b.hi[1];       // rows whose black update and output this workgroup owns
    const int j = wj(min(jraw, b.hi[1] + 1)), jm = wj(j - 1), jp = wj(j + 1);
    // rows whose lower / upper neighbour row is not the workgroup's row r - 1 / r + 1 (the halo rows; the row above the box in a partial
    // tile): 8-byte loads of that row, a plane ahead
    const bool ym_g = r == 0, yp_g = r == rw - 1 || jraw > b.hi[1];
    const int wm = ym_g ? w : w - wpr, wp = yp_g ? w : w + wpr;
    const int ih = xw * 64 + lane, iL = b.lo[0] + 2 * ih;
    // x-neighbour pairs: the adjacent lane, or (lanes 0 / 63) the last / first lane of the neighbouring wavefront of the row
    const bool hasL = lane > 0, hasR = lane < 63;
    const int wL = r * wpr + (xw == 0 ? wpr - 1 : xw - 1), wR = r * wpr + (xw == wpr - 1 ? 0 : xw + 1);
    auto parity = [&](int k) { return (b.lo[0] + j + k) & 1; };      // 0: the left cell of the pair is red (ny, nz even: wrap keeps it)
    // walls: the pair / row at a domain face (wave-uniform: wlx, whx, aty*; per lane: the first / last lane of the row)
    const bool wlx = wxl && xw == 0, whx = wxh && xw == wpr - 1;
    // open x-faces (XO: the boxes do not span the domain in x -- else a periodic x wraps inside the row as on a single box): the ghost column is loaded
    const bool olx = NBR && XO && !wxl && xw == 0, ohx = NBR && XO && !wxh && xw == wpr - 1;
    const bool atyl = wyl && j == b.lo[1], atyh = wyh && j == b.hi[1];
    auto ghost3 = [](double p0, double pin_, double pin2, double c1, double c2, double c3) { return W3 ? (p0 * c1 + pin_ * c2) + pin2 * c3 : p0 * c1 + pin_ * c2; };
    // loads: a uniform plane pointer (scalar registers) + a 32-bit byte offset per thread -- no 64-bit address registers per array and row
    typedef const __attribute__((address_space(1))) char gbyte;
    typedef double v2u __attribute__((ext_vector_type(2), aligned(8)));
    auto rowoff = [&](const FabD& f, int i, int jj) { return 8u * (unsigned)((i - f.lo[0]) + f.n[0] * (jj - f.lo[1])); };
    auto plane = [&](const FabD& f, int k, int n) { return f.gp() + (long)f.n[0] * f.n[1] * (wk(k) - f.lo[2]) + f.cs * n; };
    auto planev = [&](const FabD& f, int k, int n) {            // arrays without ghost cells (NBR: with one filled layer)
        const int kk = (wall2 && !NBR) ? min(max(k, b.lo[2]), b.hi[2]) : wk(k);
        return f.gp() + (long)f.n[0] * f.n[1] * (kk - f.lo[2]) + f.cs * n;
    };
    auto ld1 = [](const FabD::gdouble* pl, unsigned off) -> double { return *(const FabD::gdouble*)((gbyte*)pl + (size_t)off); };
    auto ldpair = [](const FabD::gdouble* pl, unsigned off) { const v2u v = *(const __attribute__((address_space(1))) v2u*)((gbyte*)pl + (size_t)off); D2 r; r.l = v.x; r.r = v.y; return r; };
    const unsigned oP = rowoff(pin, iL, j), oPy = rowoff(pin, iL, ym_g ? jm : jp);
    const unsigned oS = SG ? rowoff(S, iL, j) : 0u, oSy = SG ? rowoff(S, iL, ym_g ? jm : jp) : 0u;
    // (the right-hand side and the a-term have no ghost cells: rows / planes outside the box -- whose results nobody uses -- read the nearest valid one)
    const int jv = (wall1 && !NBR) ? min(max(j, b.lo[1]), b.hi[1]) : j;
    const unsigned oR = rowoff(rhs, iL, jv), oA = HASA ? rowoff(A, iL, jv) : 0u, oO = rowoff(pout, iL, jv);
    const bool y_g = ym_g || yp_g;
    auto ldp = [&](const FabD& f, unsigned off, int k, int n) { return ldpair(plane(f, k, n), off); };
    auto ldv = [&](const FabD& f, unsigned off, int k, int n) { return ldpair(planev(f, k, n), off); };
    auto face = [&](double s0, double s1) { return sig_scale / (0.5 * (s0 + s1)); };      // (the sum commutes: one value per face)
    // wall weights of a cell: c1 of the ghost formula at a domain face it touches, 0 elsewhere (x: per lane; y: per row; z: per plane)
    auto update = [&](double p0, double pxm, double pxp, double pym, double pyp, double pzm, double pzp, double rr, double aa,
                      double bxm, double bxp, double bym, double byp, double bzm, double bzp,
                      double cxl = 0.0, double cxh = 0.0, double cyl = 0.0, double cyh = 0.0, double czl = 0.0, double czh = 0.0) {
        return rb_update<SG, WALLS>(dhx, dhy, dhz, omega, p0, pxm, pxp, pym, pyp, pzm, pzp, rr, aa, bxm, bxp, bym, byp, bzm, bzp, cxl, cxh, cyl, cyh, czl, czh);
    };
    const double cyl0 = atyl ? bc.c1lo[1] : 0.0, cyh0 = (atyh && !atyl) ? bc.c1hi[1] : 0.0;
    constexpr bool has_a = HASA;
    const D2 Z2 = {0.0, 0.0};
    // rings: pin rows of the planes q - 1, q, q + 1; density rows q - 1 .. q + 1; rhs rows q - 1, q; new reds q - 2, q - 1, q; of the red
    // cell's face coefficients: the pair's own face and the face towards the next pair (plane q - 1), the face above (planes q - 2, q - 1).
    // PN / SN / RN / AN / YNN: the rows in flight.  An iteration first moves them into the rings (the only place that waits for them: they
    // were issued a whole iteration earlier) and then issues the next loads into the same registers -- a ring rotation at the END of the
    // iteration would wait for the loads it has just issued.
    D2 Pm = Z2, Pc, Pp, PN, Sm = Z2, Sc = Z2, Sp = Z2, SN = Z2, Rm = Z2, Rc, RN, Am = Z2, Ac = Z2, AN = Z2;
    double RNa = 0.0, RNm = 0.0, RNc = 0.0;
    double bnear_m = bu.v[0], bfar_m = bu.v[0], bzp_a = bu.v[2], bzp_m = bu.v[2], bzm_c = bu.v[2];
    // halo rows: old phi and density of the red cell's y-neighbour outside the tile, plane qq
    struct YN { double p, s; };
    auto yload = [&](int qq, int parq) {
        YN y = {0.0, 0.0};
        if (y_g) {
            const unsigned dr = 8u * (unsigned)parq;
            if (!zero) y.p = ld1(plane(pin, qq, comp), oPy + dr);
            if (SG) y.s = ld1(plane(S, qq, sig_comp), oSy + dr);
        }
        return y;
    };
    // walls in x (density form): the density beyond the face, loaded by the first / last lane of the row a plane ahead (planes q - 1, q, in flight)
    const bool xs_on = !(IAMRX_RBW_EXP & 1) && SG && (((wlx || olx) && !hasL) || ((whx || ohx) && !hasR));
    const unsigned oSx = xs_on ? rowoff(S, ((wlx || olx) && !hasL) ? b.lo[0] - 1 : b.hi[0] + 1, j) : 0u;
    double xs_m = 0.0, xs_c = 0.0, xsN = 0.0;
    auto xsload = [&](int qq) { return xs_on ? ld1(plane(S, qq, sig_comp), oSx) : 0.0; };
    // open x-faces: phi of the ghost column (old where it is black, k_abec_rb_ghost's new value where it is red), planes q - 1, q, in flight
    const bool xp_on = NBR && XO && ((olx && !hasL) || (ohx && !hasR));
    const unsigned oPx = xp_on ? rowoff(pin, (olx && !hasL) ? b.lo[0] - 1 : b.hi[0] + 1, j) : 0u;
    double xp_m = 0.0, xp_c = 0.0, xpN = 0.0;
    auto xpload = [&](int qq) { return xp_on ? ld1(plane(pin, qq, comp), oPx) : 0.0; };
    // W3: the rows two inside a y-face of the box, for the pairs of the face rows (planes q - 1, q, in flight)
    const bool y2_on = W3 && (atyl || atyh);
    const unsigned oP2 = y2_on ? rowoff(pin, iL, atyl ? j + 2 : j - 2) : 0u;
    D2 Y2m = Z2, Y2c = Z2, Y2N = Z2;
    auto y2load = [&](int qq) { return (y2_on && !zero) ? ldp(pin, oP2, qq, comp) : Z2; };
    // zero start: phi is not read -- but for the ghost rows / planes of an open face, which hold k_abec_rb_ghost's reds (and zeros)
    auto pload = [&](int kk) { return (!zero || (NBR && (row_out || kk < b.lo[2] || kk > b.hi[2]))) ? ldp(pin, oP, kk, comp) : Z2; };
    double bn_prev = 0.0;
    D2 CN = Z2;             // ACC: the solution's pair of the plane the next `put` writes
    auto cload = [&](int k) { return ldpair(pout.gp() + (long)pout.n[0] * pout.n[1] * (k - pout.lo[2]) + pout.cs * comp, oO); };
    auto put = [&](int k, int park, double bn, double rn) {          // park 1: the left cell of the pair is the black one
        v2u o;
        if (park) { o.x = bn; o.y = rn; } else { o.x = rn; o.y = bn; }
        if (ACC) { o.x = CN.l + o.x; o.y = CN.r + o.y; }
        typedef __attribute__((address_space(1))) char gwbyte;
        *(__attribute__((address_space(1))) v2u*)((gwbyte*)(pout.gp() + (long)pout.n[0] * pout.n[1] * (k - pout.lo[2]) + pout.cs * comp) + (size_t)oO) = o;
    };
    int q = k0 - 1;
    // (the state an iteration q - 1 would have left)
    Pc = pload(q - 1); Pp = pload(q); PN = pload(q + 1);
    if (SG) { Sc = ldp(S, oS, q - 1, sig_comp); Sp = ldp(S, oS, q, sig_comp); SN = ldp(S, oS, q + 1, sig_comp); }
    Rc = ldv(rhs, oR, q - 1, comp); RN = ldv(rhs, oR, q, comp);
    if (has_a) { Ac = ldv(A, oA, q - 1, 0); AN = ldv(A, oA, q, 0); }
    YN yn, YNN = yload(q, parity(q));
    if (WALLS || (NBR && XO)) { xs_c = xsload(q - 1); xsN = xsload(q); }
    if (NBR && XO) { xp_c = xpload(q - 1); xpN = xpload(q); }
    if (W3) { Y2c = y2load(q - 1); Y2N = y2load(q); }
    {   // black cells of the first plane
        const int par = parity(q);
        BPH[0][w][lane] = par ? Pp.l : Pp.r;
        if (SG) BSG[0][w][lane] = par ? Sp.l : Sp.r;
        __syncthreads();
    }
    // one plane; PAR = parity(q) as a type (0: the left cell of the pair is the red one of plane q): the selections of pair entries and
    // lane-shift directions are static -- the march below runs two planes per trip, in the order the row's parity asks for
    auto iter = [&](auto PARC, const int q) {
        constexpr int par = decltype(PARC)::value;
        // ---- the rows in flight enter the rings; loads of the next plane
        // (rb_take: an explicit register move, so that the copy stays HERE and the new load lands in the register just vacated)
        Pm = Pc; Pc = Pp; if (!zero || NBR) { Pp.l = rb_take(PN.l); Pp.r = rb_take(PN.r); }
        if (SG) { Sm = Sc; Sc = Sp; Sp.l = rb_take(SN.l); Sp.r = rb_take(SN.r); }
        Rm = Rc; Rc.l = rb_take(RN.l); Rc.r = rb_take(RN.r);
        if (has_a) { Am = Ac; Ac.l = rb_take(AN.l); Ac.r = rb_take(AN.r); }
        yn = YNN;
        if (y_g) { if (!zero) yn.p = rb_take(YNN.p); if (SG) yn.s = rb_take(YNN.s); }
        if ((WALLS || (NBR && XO)) && SG) { xs_m = xs_c; xs_c = xs_on ? rb_take(xsN) : 0.0; }
        if (NBR && XO) { xp_m = xp_c; xp_c = xp_on ? rb_take(xpN) : 0.0; }
        if (W3) { Y2m = Y2c; if (y2_on && !zero) { Y2c.l = rb_take(Y2N.l); Y2c.r = rb_take(Y2N.r); } }
        // the output of the previous iteration's black update (plane q - 2), BEFORE the loads: the wait for the loads at the top of the
        // next iteration then covers nothing younger than a whole iteration (a store behind them would be waited for as well)
        if (owner && q - 2 >= k0) put(q - 2, par, bn_prev, RNm);
        if (q <= kend) {
            if (NBR) PN = pload(q + 2); else if (!zero) PN = ldp(pin, oP, q + 2, comp);
            if (SG) SN = ldp(S, oS, q + 2, sig_comp);
            RN = ldv(rhs, oR, q + 1, comp);
            if (has_a) AN = ldv(A, oA, q + 1, 0);
            YNN = yload(q + 1, 1 - par);
            if ((WALLS || (NBR && XO)) && SG) xsN = xsload(q + 1);
            if (NBR && XO) xpN = xpload(q + 1);
            if (W3 && y2_on && !zero) Y2N = y2load(q + 1);
        }
        if (ACC && owner && q - 1 >= k0 && q - 1 <= kend) CN = cload(q - 1);        // (consumed by the put at the top of the next iteration)
        // ---- red update of plane q
        const int bq = (q - k0 + 1) & 1;
        const int slot = (q - k0 + 3) % 3, slotk = (q - k0 + 2) % 3;      // ring slots of the planes q and q - 1
        double bnear_c = bu.v[0], bfar_c = bu.v[0], bzp_c = bu.v[2];
        {
            // the black cell of the pair next to the red one: par 0: red = left, the pair to the left; par 1: the pair to the right
            double nb = par == 0 ? rb_lane_lo(Pc.r) : rb_lane_hi(Pc.l), nbs = 0.0;
            if (SG) nbs = par == 0 ? rb_lane_lo(Sc.r) : rb_lane_hi(Sc.l);
            const double p0 = par ? Pc.r : Pc.l;
            // (W3: the old red two cells inside an x-face = the red cell of the next pair)
            const double x2 = W3 ? (par == 0 ? rb_lane_hi(Pc.l) : rb_lane_lo(Pc.r)) : 0.0;
            double cxl = 0.0, cxh = 0.0, czl = 0.0, czh = 0.0;
            if (par == 0 ? !hasL : !hasR) {
                if (WALLS && !(IAMRX_RBW_EXP & 8) && (par == 0 ? wlx : whx)) {           // the red cell sits at an x-face of the domain
                    nb = ghost3(p0, par == 0 ? Pc.r : Pc.l, x2, par == 0 ? bc.c1lo[0] : bc.c1hi[0], par == 0 ? bc.c2lo[0] : bc.c2hi[0], par == 0 ? bc.c3lo[0] : bc.c3hi[0]);
                    if (SG && !(IAMRX_RBW_EXP & 64)) nbs = xs_c;
                    if (!(IAMRX_RBW_EXP & 32)) { if (par == 0) cxl = bc.c1lo[0]; else cxh = bc.c1hi[0]; }
                } else if (NBR && XO && (par == 0 ? olx : ohx)) {      // ... at an open x-face: the old black value of the ghost column
                    nb = zero ? 0.0 : xp_c;
                    if (SG) nbs = xs_c;
                } else {
                nb = BPH[bq][par == 0 ? wL : wR][par == 0 ? 63 : 0];
                if (SG) nbs = BSG[bq][par == 0 ? wL : wR][par == 0 ? 63 : 0];
                }
            }
            double pym = BPH[bq][wm][lane], pyp = BPH[bq][wp][lane], sym = 0.0, syp = 0.0;
            if (SG) { sym = BSG[bq][wm][lane]; syp = BSG[bq][wp][lane]; }
            if (ym_g) { pym = yn.p; sym = yn.s; }
            if (yp_g) { pyp = yn.p; syp = yn.s; }
            double pzm = par ? Pm.r : Pm.l, pzp = par ? Pp.r : Pp.l;
            if (WALLS && !(IAMRX_RBW_EXP & 4)) {
                const double y2 = W3 ? (par ? Y2c.r : Y2c.l) : 0.0;
                if (atyl) pym = ghost3(p0, pyp, y2, bc.c1lo[1], bc.c2lo[1], bc.c3lo[1]);
                else if (atyh) pyp = ghost3(p0, pym, y2, bc.c1hi[1], bc.c2hi[1], bc.c3hi[1]);
                if (wzl && q == b.lo[2]) {
                    // (W3: the old red of plane q + 2 is the row in flight: one wait per tile of the first chunk)
                    const double z2 = W3 ? (par ? PN.r : PN.l) : 0.0;
                    pzm = ghost3(p0, pzp, z2, bc.c1lo[2], bc.c2lo[2], bc.c3lo[2]); czl = bc.c1lo[2];
                } else if (wzh && q == b.hi[2]) {
                    double z2 = 0.0;
                    if (W3 && !zero) { const D2 t2 = ldp(pin, oP, q - 2, comp); z2 = par ? t2.r : t2.l; }
                    pzp = ghost3(p0, pzm, z2, bc.c1hi[2], bc.c2hi[2], bc.c3hi[2]); czh = bc.c1hi[2];
                }
            }
            const double pxm = par == 0 ? nb : Pc.l, pxp = par == 0 ? Pc.r : nb;
            const double rr = par ? Rc.r : Rc.l;
            const double aa = has_a ? alpha * (par ? Ac.r : Ac.l) : 0.0;
            double bxm = bu.v[0], bxp = bu.v[0], bym = bu.v[1], byp = bu.v[1];
            if (SG) {
                const double s0 = par ? Sc.r : Sc.l;
                bnear_c = face(s0, par ? Sc.l : Sc.r); bfar_c = face(s0, nbs);
                bxm = par == 0 ? bfar_c : bnear_c; bxp = par == 0 ? bnear_c : bfar_c;
                bym = face(sym, s0); byp = face(s0, syp);
                bzm_c = face(par ? Sm.r : Sm.l, s0); bzp_c = face(s0, par ? Sp.r : Sp.l);
            }
            RNa = RNm; RNm = RNc;
            RNc = update(p0, pxm, pxp, pym, pyp, pzm, pzp, rr, aa, bxm, bxp, bym, byp, bzm_c, bzp_c, cxl, cxh, (IAMRX_RBW_EXP & 4) ? 0.0 : cyl0, (IAMRX_RBW_EXP & 4) ? 0.0 : cyh0, czl, czh);
            if (NBR && (row_out || q < b.lo[2] || q > b.hi[2])) RNc = p0;      // a ghost row / plane: already updated (k_abec_rb_ghost)
            if (SG) {
                YM[slot][w][lane] = bym; YP[slot][w][lane] = byp;
                if (!hasL) XF[slot][w][0] = bfar_c;
                if (!hasR) XF[slot][w][1] = bfar_c;
            }
        }
        asm volatile("" : "+v"(RNc));          // (ends the scheduling region: the two updates of an iteration are not interleaved)
        RED[slot][w][lane] = RNc;
        BPH[bq ^ 1][w][lane] = par ? Pp.r : Pp.l;                // the black cells of plane q + 1 (its parity is 1 - par)
        if (SG) BSG[bq ^ 1][w][lane] = par ? Sp.r : Sp.l;
        __syncthreads();
        // ---- black update of plane k = q - 1 and its output
        const int k = q - 1;
        if (k >= k0 && owner) {
            constexpr int park = 1 - par;                                    // parity(k): 1: the left cell of the pair is black
            // (the lane shifts below run under the row mask `owner`: every lane of a wavefront belongs to one row)
            double nb = park == 1 ? rb_lane_lo(RNm) : rb_lane_hi(RNm), nbf = bu.v[0];
            if (SG) nbf = park == 1 ? rb_lane_lo(bfar_m) : rb_lane_hi(bfar_m);
            const double p0 = park ? Pm.l : Pm.r, rr = park ? Rm.l : Rm.r;
            // (W3: the old black two cells inside an x-face = the black cell of the next pair)
            const double x2 = W3 ? (park == 1 ? rb_lane_hi(Pm.l) : rb_lane_lo(Pm.r)) : 0.0;
            double cxl = 0.0, cxh = 0.0, czl = 0.0, czh = 0.0;
            if (park == 1 ? !hasL : !hasR) {
                if (WALLS && !(IAMRX_RBW_EXP & 8) && (park == 1 ? wlx : whx)) {          // the black cell sits at an x-face: ghost from its own value and the new red beside it
                    nb = ghost3(p0, RNm, x2, park == 1 ? bc.c1lo[0] : bc.c1hi[0], park == 1 ? bc.c2lo[0] : bc.c2hi[0], park == 1 ? bc.c3lo[0] : bc.c3hi[0]);
                    if (SG && !(IAMRX_RBW_EXP & 16)) nbf = face(park ? Sm.l : Sm.r, xs_m);
                    if (!(IAMRX_RBW_EXP & 32)) { if (park == 1) cxl = bc.c1lo[0]; else cxh = bc.c1hi[0]; }
                } else if (NBR && XO && (park == 1 ? olx : ohx)) {     // ... at an open x-face: the new red value of the ghost column
                    nb = xp_m;
                    if (SG) nbf = face(park ? Sm.l : Sm.r, xs_m);
                } else {
                nb = RED[slotk][park == 1 ? wL : wR][park == 1 ? 63 : 0];
                if (SG) nbf = XF[slotk][park == 1 ? wL : wR][park == 1 ? 1 : 0];
                }
            }
            const double pxm = park == 1 ? nb : RNm, pxp = park == 1 ? RNm : nb;
            double pym = RED[slotk][w - wpr][lane], pyp = RED[slotk][w + wpr][lane], pzm = RNa, pzp = RNc;
            if (WALLS && !(IAMRX_RBW_EXP & 4)) {
                const double y2 = W3 ? (park ? Y2m.l : Y2m.r) : 0.0;
                if (atyl) pym = ghost3(p0, pyp, y2, bc.c1lo[1], bc.c2lo[1], bc.c3lo[1]);
                else if (atyh) pyp = ghost3(p0, pym, y2, bc.c1hi[1], bc.c2hi[1], bc.c3hi[1]);
                if (wzl && k == b.lo[2]) {
                    const double z2 = W3 ? (park ? Pp.l : Pp.r) : 0.0;          // the old black of plane k + 2 = q + 1
                    pzm = ghost3(p0, pzp, z2, bc.c1lo[2], bc.c2lo[2], bc.c3lo[2]); czl = bc.c1lo[2];
                } else if (wzh && k == b.hi[2]) {
                    double z2 = 0.0;
                    if (W3 && !zero) { const D2 t2 = ldp(pin, oP, k - 2, comp); z2 = park ? t2.l : t2.r; }
                    pzp = ghost3(p0, pzm, z2, bc.c1hi[2], bc.c2hi[2], bc.c3hi[2]); czh = bc.c1hi[2];
                }
            }
            double bxm = bu.v[0], bxp = bu.v[0], bym = bu.v[1], byp = bu.v[1], bzm = bu.v[2], bzp = bu.v[2];
            if (SG) {
                bxm = park == 1 ? nbf : bnear_m; bxp = park == 1 ? bnear_m : nbf;
                bym = YP[slotk][w - wpr][lane]; byp = YM[slotk][w + wpr][lane];
                bzm = bzp_a; bzp = bzm_c;
            }
            const double aa = has_a ? alpha * (park ? Am.l : Am.r) : 0.0;
            const double bn = update(p0, pxm, pxp, pym, pyp, pzm, pzp, rr, aa, bxm, bxp, bym, byp, bzm, bzp, cxl, cxh, (IAMRX_RBW_EXP & 4) ? 0.0 : cyl0, (IAMRX_RBW_EXP & 4) ? 0.0 : cyh0, czl, czh);
            bn_prev = bn;
        }
        if (SG) { bnear_m = bnear_c; bfar_m = bfar_c; bzp_a = bzp_m; bzp_m = bzp_c; }
    };
    typedef std::integral_constant<int, 0> Par0;
    typedef std::integral_constant<int, 1> Par1;
    if (parity(q) == 0) for (; q <= kend + 1; ++q) { iter(Par0{}, q); if (++q > kend + 1) break; iter(Par1{}, q); }
    else                for (; q <= kend + 1; ++q) { iter(Par1{}, q); if (++q > kend + 1) break; iter(Par0{}, q); }
    q = kend + 2;
    if (owner && q - 2 >= k0) put(q - 2, parity(q), bn_prev, RNm);
#endif
}

// k_abec_rb_ghost: in front of k_abec_gsrb_rb<.., NBR> -- the red update of the RED ghost cells next to the open faces of every box (the
// cells one layer outside the box whose values the black update of the box's surface cells reads), in place in pin.  Nobody else reads the
// old value of a red cell (a red update reads the cell itself and black neighbours), and the neighbours of a ghost cell are phi's two filled
// ghost layers.  The box behind the face computes the same update of the same cell from the same doubles with the same expression
// (rb_update, faces per rb_face): the value is the one a ghost fill between the two colour passes would have delivered.  A ghost cell that
// touches a domain wall in another direction applies the wall's ghost formula like the cells inside.  zero: pin counts as identically zero
// and is not read; the black ghost cells of the open faces are then zeroed here (the sweep kernel reads them).
// grid: x = 256-cell pieces of a face, y = 6 * local box + 2 * direction + side
template <int BMODE, bool HASA, bool WALLS>
__global__ void __launch_bounds__(256) k_abec_rb_ghost(const BoxD* __restrict__ boxes, const FabD* __restrict__ pint, const FabD* __restrict__ rhst,
    const FabD* __restrict__ At, const FabD* __restrict__ St, double alpha, double dhx, double dhy, double dhz, double omega, int sig_comp, double sig_scale,
    BUni bu, int zero, int comp, RbBC bc, int xo)
{
    constexpr bool SG = BMODE == 1;
    const int fab = (int)blockIdx.y / 6, face_id = (int)blockIdx.y % 6, d = face_id >> 1, side = face_id & 1;
    const BoxD b = boxes[fab];
    if (d == 0 && !xo) return;                   // rows that span the domain: a periodic x wraps inside the sweep kernel
    if (WALLS && !bc.per[d] && (side == 0 ? b.lo[d] == bc.dlo[d] : b.hi[d] == bc.dhi[d])) return;      // a wall: nothing behind it
    const int d1 = d == 0 ? 1 : 0, d2 = d == 2 ? 1 : 2;
    const int n1 = b.len(d1), n2 = b.len(d2);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n1 * n2) return;
    int c[3];
    c[d] = side == 0 ? b.lo[d] - 1 : b.hi[d] + 1;
    c[d1] = b.lo[d1] + (int)(idx % n1);
    c[d2] = b.lo[d2] + (int)(idx / n1);
    const int i = c[0], j = c[1], k = c[2];
    const FabD pin = pint[fab];
    if ((i + j + k) & 1) { if (zero) pin(i, j, k, comp) = 0.0; return; }
    const FabD rhs = rhst[fab];
    auto P = [&](int ii, int jj, int kk) -> double { return zero ? 0.0 : (double)pin(ii, jj, kk, comp); };
    auto ghost = [](double p0, double pin_, double c1, double c2) { return p0 * c1 + pin_ * c2; };
    const double p0 = P(i, j, k);
    double pxm = P(i - 1, j, k), pxp = P(i + 1, j, k), pym = P(i, j - 1, k), pyp = P(i, j + 1, k), pzm = P(i, j, k - 1), pzp = P(i, j, k + 1);
    double cw[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // wall weights (lo, hi) per direction, see rb_update
    if (WALLS) {
        if (!bc.per[0]) { if (i == bc.dlo[0]) { pxm = ghost(p0, pxp, bc.c1lo[0], bc.c2lo[0]); cw[0] = bc.c1lo[0]; } else if (i == bc.dhi[0]) { pxp = ghost(p0, pxm, bc.c1hi[0], bc.c2hi[0]); cw[1] = bc.c1hi[0]; } }
        if (!bc.per[1]) { if (j == bc.dlo[1]) { pym = ghost(p0, pyp, bc.c1lo[1], bc.c2lo[1]); cw[2] = bc.c1lo[1]; } else if (j == bc.dhi[1]) { pyp = ghost(p0, pym, bc.c1hi[1], bc.c2hi[1]); cw[3] = bc.c1hi[1]; } }
        if (!bc.per[2]) { if (k == bc.dlo[2]) { pzm = ghost(p0, pzp, bc.c1lo[2], bc.c2lo[2]); cw[4] = bc.c1lo[2]; } else if (k == bc.dhi[2]) { pzp = ghost(p0, pzm, bc.c1hi[2], bc.c2hi[2]); cw[5] = bc.c1hi[2]; } }
    }
    double bxm = bu.v[0], bxp = bu.v[0], bym = bu.v[1], byp = bu.v[1], bzm = bu.v[2], bzp = bu.v[2];
    if (SG) {
        const FabD S = St[fab];
        auto fc = [&](double s0, double s1) { return sig_scale / (0.5 * (s0 + s1)); };
        const double s0 = S(i, j, k, sig_comp);
        bxm = fc(s0, S(i - 1, j, k, sig_comp)); bxp = fc(s0, S(i + 1, j, k, sig_comp));
        bym = fc(S(i, j - 1, k, sig_comp), s0); byp = fc(s0, S(i, j + 1, k, sig_comp));
        bzm = fc(S(i, j, k - 1, sig_comp), s0); bzp = fc(s0, S(i, j, k + 1, sig_comp));
    }
    double aa = 0.0;
    if (HASA) { const FabD A = At[fab]; aa = alpha * A(i, j, k, 0); }
    pin(i, j, k, comp) = rb_update<SG, WALLS>(dhx, dhy, dhz, omega, p0, pxm, pxp, pym, pyp, pzm, pzp, rhs(i, j, k, comp), aa, bxm, bxp, bym, byp, bzm, bzp,
                                              cw[0], cw[1], cw[2], cw[3], cw[4], cw[5]);
}

// the ghost formula of one component's boundary conditions as the sweep kernel applies it; false: a condition it does not take
static bool rb_make_bc(const Geometry& g, const DomainBC& bc, RbBC& r)
{
    for (int d = 0; d < 3; ++d) {
        r.per[d] = g.periodic[d] ? 1 : 0;
        r.c1lo[d] = r.c2lo[d] = r.c1hi[d] = r.c2hi[d] = 0.0;
        if (g.periodic[d]) continue;
        for (int side = 0; side < 2; ++side) {
            const int t = side == 0 ? bc.lo[d] : bc.hi[d];
            double c1, c2 = 0.0;
            if (t == lo_neumann) c1 = 1.0;
            else if (t == lo_reflect_odd) c1 = -1.0;
            else if (t == lo_dirichlet) {
                double c[4]; int NX;
                dirichlet_coefs(g.domain.len(d), bc.maxorder, c, NX);
                if (NX < 2 || NX > 3) return false;
                c1 = c[1]; c2 = c[2];
            } else return false;
            (side == 0 ? r.c1lo[d] : r.c1hi[d]) = c1;
            (side == 0 ? r.c2lo[d] : r.c2hi[d]) = c2;
        }
    }
    return true;
}

// bcs: the level's boundary conditions (nbc sets: one per component, or one for all); null: fully periodic levels only.
// IAMRX_GSRB_RB_WALLS (1): 0 = index-wrap levels only.
// (decided from the level's global box list: every rank takes the same path -- ADVICE round 4; a rank without the box launches nothing)
bool abec_gsrb_rb_ok(const Geometry& g, const AbecCoef& c, const MultiFab& phi, int nbc, const DomainBC* bcs)
{
    if (tune("GSRB_RB", 1) == 0 || tune("ABEC_SIG", 1) == 0 || tune("PERIODIC_WRAP", 1) == 0) return false;
    const Layout& l = *phi.layout;
    if (l.boxes.size() != 1) return false;             // (decided on the GLOBAL box list: every rank takes the same path, a rank that does not own the box launches nothing)
    const BoxD& b = l.boxes[0];
    bool walls = false;
    for (int d = 0; d < 3; ++d) {
        if (b.lo[d] != g.domain.lo[d] || b.hi[d] != g.domain.hi[d] || b.len(d) < 4) return false;
        if (!g.periodic[d]) walls = true;
    }
    if (walls) {
        if (!bcs || nbc < 1 || tune("GSRB_RB_WALLS", 1) == 0) return false;
        for (int n = 0; n < phi.ncomp; ++n) { RbBC r; if (!rb_make_bc(g, bcs[n < nbc ? n : 0], r)) return false; }
    }
    const int nx = b.len(0);
    if (nx != 128 && nx != 256) return false;                      // whole rows in 1 or 2 wavefronts of a workgroup
    if ((b.len(1) & 1) || (b.len(2) & 1) || b.len(1) < 16 || b.len(2) < 16) return false;
    if (phi.ngrow < 1 || c.b[0]->ncomp != 1) return false;
    if (c.sig) return phi.ncomp == 1 && c.sig->ngrow >= 1 && !(c.a && c.alpha != 0.0);
    return c.b_uniform != 0;
}

// The sweep on a REFINED level that is one box strictly inside its domain (k_abec_gsrb_rb<.., WALLS, .., W3>): every face of the box is a
// coarse/fine face, whose homogeneous ghost value (k_cf_fill, correction form) is the three-weight formula of the level's CfTab -- the
// kernel evaluates it on the values at hand and reads no ghost cell of phi.  IAMRX_GSRB_RB_CF (1): 0 = colour passes (k_abec_gsrb2<.., CF>).
bool abec_gsrb_rb_cf_ok(const Geometry& g, const AbecCoef& c, const MultiFab& phi)
{
    if (tune("GSRB_RB", 1) == 0 || tune("ABEC_SIG", 1) == 0 || tune("GSRB_RB_CF", 1) == 0) return false;
    const Layout& l = *phi.layout;
    if (l.boxes.size() != 1) return false;             // (decided on the GLOBAL box list: every rank takes the same path, a rank that does not own the box launches nothing)
    const BoxD& b = l.boxes[0];
    for (int d = 0; d < 3; ++d) if (!(b.lo[d] > g.domain.lo[d] && b.hi[d] < g.domain.hi[d])) return false;
    const int nx = b.len(0);
    if (nx != 128 && nx != 256) return false;
    if ((b.len(1) & 1) || (b.len(2) & 1) || b.len(1) < 16 || b.len(2) < 16) return false;
    if (phi.ngrow < 1 || c.b[0]->ncomp != 1) return false;
    if (c.sig) return phi.ncomp == 1 && c.sig->ngrow >= 1 && !(c.a && c.alpha != 0.0);
    return c.b_uniform != 0;
}

// The sweep on a level of SEVERAL boxes that covers its domain (k_abec_gsrb_rb<.., NBR>: a chopped level, the boxes of a sharded level):
// what the level must look like -- rows of 128 or 256 cells in every box, no coarse/fine faces (the caller knows), wall conditions the
// kernel has a ghost formula for.  The arrays: phi (both buffers) with two ghost layers, the density with two, rhs and the a-term with one
// (abec_gsrb_rb_nbr asserts them).  IAMRX_GSRB_RB_NBR (1): 0 = colour passes with a ghost fill in front of each.
bool abec_gsrb_rb_nbr_level_ok(const Geometry& g, const Layout& l, int ncomp, bool sig_form, bool has_a, int nbc, const DomainBC* bcs)
{
    if (tune("GSRB_RB", 1) == 0 || tune("ABEC_SIG", 1) == 0 || tune("GSRB_RB_NBR", 1) == 0) return false;
    if (l.boxes.size() < 2 || l.total_cells() != g.domain.npts()) return false;
    const int nx = l.boxes[0].len(0);
    if (nx != 128 && nx != 256) return false;
    for (const BoxD& b : l.boxes) if (b.len(0) != nx || b.len(1) < 16 || b.len(2) < 16) return false;
    bool walls = false;
    for (int d = 0; d < 3; ++d) if (!g.periodic[d]) walls = true;
    if (walls) {
        if (!bcs || nbc < 1 || tune("GSRB_RB_WALLS", 1) == 0) return false;
        for (int n = 0; n < ncomp; ++n) { RbBC r; if (!rb_make_bc(g, bcs[n < nbc ? n : 0], r)) return false; }
    }
    if (sig_form) return ncomp == 1 && !has_a;
    return true;
}

bool abec_gsrb_rb_nbr_ok(const Geometry& g, const AbecCoef& c, const MultiFab& phi, const MultiFab& rhs, int nbc, const DomainBC* bcs)
{
    const bool has_a = c.a && c.alpha != 0.0;
    if (!abec_gsrb_rb_nbr_level_ok(g, *phi.layout, phi.ncomp, c.sig != nullptr, has_a, nbc, bcs)) return false;
    if (phi.ngrow < 2 || rhs.ngrow < 1 || c.b[0]->ncomp != 1) return false;
    if (c.sig) return c.sig->ngrow >= 2;
    return c.b_uniform != 0 && (!has_a || c.a->ngrow >= 1);
}

// one red + black sweep pin -> pout (pin != pout); zero: pin is identically zero and is not read
template <int NW>
static void abec_gsrb_rb_nw(const Geometry& g, const AbecCoef& c, const MultiFab& pin, MultiFab& pout, const MultiFab& rhs, double omega, bool zero,
                            const DomainBC* bcs, int nbc, const CfTab* cf, bool acc)
{
    const bool walls = cf != nullptr || !(g.periodic[0] && g.periodic[1] && g.periodic[2]);
    auto& ctx = Context::get();
    const Layout& l = *pin.layout;
    const BoxD b = l.boxes[l.local[0]];
    const int wpr = b.len(0) / 128, rw = NW / wpr;
    const int nty = (b.len(1) + (rw - 2) - 1) / (rw - 2);
    // z-chunks: one round of workgroups (one 1024-thread workgroup per CU)
    int nch = std::max(1, (int)(tune("GSRB_RB_SLOTS", 256) / nty));
    int tz = std::max(8, (b.len(2) + nch - 1) / nch);
    nch = (b.len(2) + tz - 1) / tz;
    const int ntile = nty * nch, xcd_chunk = tune("GSRB_RB_XCD", 1) != 0 ? (ntile + 7) / 8 : 0, nwg = xcd_chunk > 0 ? 8 * xcd_chunk : ntile;
    const double dhx = c.beta / (g.dx[0] * g.dx[0]), dhy = c.beta / (g.dx[1] * g.dx[1]), dhz = c.beta / (g.dx[2] * g.dx[2]);
    const bool has_a = c.a && c.alpha != 0.0;
    const FabD P = pin.h_tab[0], O = pout.h_tab[0], R = rhs.h_tab[0], Af = has_a ? c.a->h_tab[0] : P, Sf = c.sig ? c.sig->h_tab[0] : P;
    // (the in-step probe of bench.py times the plain sweep: the last sweep of a cycle also carries the `sol += cor` pass, ACC)
    const bool rec = !acc && pin.ncomp == 1 && kernel_probe_begin(PROBE_ABEC_GSRB, (long)l.max_len[0] * l.max_len[1] * l.max_len[2]);
    for (int n = 0; n < pin.ncomp; ++n) {
        BUni bn;
        for (int d = 0; d < 3; ++d) bn.v[d] = c.bu[d] * ((c.tensor_eta && n == d) ? 4.0 / 3.0 : 1.0);
        RbBC rbc;
        if (cf) {
            // every face a coarse/fine face: k_cf_fill's weights (NX = min(len + 1, maxorder) points, the first of them the boundary value: zero here)
            for (int d = 0; d < 3; ++d) {
                const int NX = std::min(b.len(d) + 1, cf->maxorder);
                const double* w = cf->c[d][NX - 2];
                rbc.per[d] = 0;
                rbc.c1lo[d] = rbc.c1hi[d] = w[1];
                rbc.c2lo[d] = rbc.c2hi[d] = NX > 2 ? w[2] : 0.0;
                rbc.c3lo[d] = rbc.c3hi[d] = NX > 3 ? w[3] : 0.0;
                rbc.dlo[d] = b.lo[d]; rbc.dhi[d] = b.hi[d];
            }
        } else
        if (walls) rb_make_bc(g, bcs[n < nbc ? n : 0], rbc);
#define IAMRX_RBL(M, HA, WL, T3, AC, SC, SS) hipLaunchKernelGGL((k_abec_gsrb_rb<M, NW, HA, WL, false, false, T3, AC>), dim3((unsigned)nwg), dim3(64 * NW), 0, ctx.stream, b, P, O, R, Af, Sf, \
                                                      c.alpha, dhx, dhy, dhz, omega, SC, SS, bn, wpr, tz, nty, zero ? 1 : 0, n, xcd_chunk, rbc)
#define IAMRX_RB(M, HA, WL, SC, SS) do { if (acc) IAMRX_RBL(M, HA, WL, false, true, SC, SS); else IAMRX_RBL(M, HA, WL, false, false, SC, SS); } while (0)
#define IAMRX_RB3(M, HA, SC, SS) do { if (acc) IAMRX_RBL(M, HA, true, true, true, SC, SS); else IAMRX_RBL(M, HA, true, true, false, SC, SS); } while (0)
        if (cf) {
            if (c.sig) IAMRX_RB3(1, false, c.sig_comp, c.sig_scale);
            else if (has_a) IAMRX_RB3(2, true, 0, 1.0);
            else IAMRX_RB3(2, false, 0, 1.0);
        } else
        if (walls) {
            if (c.sig) IAMRX_RB(1, false, true, c.sig_comp, c.sig_scale);
            else if (has_a) IAMRX_RB(2, true, true, 0, 1.0);
            else IAMRX_RB(2, false, true, 0, 1.0);
        } else {
            if (c.sig) IAMRX_RB(1, false, false, c.sig_comp, c.sig_scale);
            else if (has_a) IAMRX_RB(2, true, false, 0, 1.0);
            else IAMRX_RB(2, false, false, 0, 1.0);
        }
#undef IAMRX_RB
#undef IAMRX_RB3
#undef IAMRX_RBL
    }
    if (rec) kernel_probe_end(PROBE_ABEC_GSRB);
}
void abec_gsrb_rb(const Geometry& g, const AbecCoef& c, const MultiFab& pin, MultiFab& pout, const MultiFab& rhs, double omega, bool zero,
                  const DomainBC* bcs, int nbc, const CfTab* cf, bool acc)
{
    IAMRX_ASSERT((cf ? abec_gsrb_rb_cf_ok(g, c, pin) : abec_gsrb_rb_ok(g, c, pin, nbc, bcs)) && pin.d_tab != pout.d_tab && pout.ngrow >= 1 && rhs.ncomp == pin.ncomp);
    if (pin.nlocal() == 0) return;
    // The density form on a domain with walls needs more than the 128 VGPRs a 1024-thread workgroup leaves a thread (72 bytes of scratch per
    // lane, 210 us per 256^3 sweep): 12 wavefronts (768 threads: 168 VGPRs) trade two of the eight rows of a tile for a spill-free loop
    const bool walls = cf != nullptr || !(g.periodic[0] && g.periodic[1] && g.periodic[2]);
    const int wpr = pin.layout->boxes[pin.layout->local[0]].len(0) / 128;
    if (walls && c.sig && 12 % wpr == 0 && 12 / wpr >= 3 && tune("GSRB_RB_NW12", 1) != 0) abec_gsrb_rb_nw<12>(g, c, pin, pout, rhs, omega, zero, bcs, nbc, cf, acc);
    else abec_gsrb_rb_nw<16>(g, c, pin, pout, rhs, omega, zero, bcs, nbc, cf, acc);
}

// the sweep of this level can be issued in two parts -- the tiles that read no ghost cell (sel 1) and the others (sel 2, behind the exchange):
// rows that span the domain (no ghost columns) and boxes tall enough for tiles in the middle
bool abec_gsrb_rb_nbr_splits(const Geometry& g, const Layout& l)
{
    return l.max_len[0] == g.domain.len(0) && l.max_len[1] >= 48 && l.max_len[2] >= 32;
}

// the same on a level of several boxes (see k_abec_gsrb_rb<.., NBR>): pin's two ghost layers, rhs's (and the a-term's) one and the density's
// two are filled by the caller -- neighbour boxes and periodic images; nothing is read beyond a domain wall but the density's first layer.
// pin's red ghost cells next to open faces are overwritten (k_abec_rb_ghost); zero: pin's valid cells are not read, those ghost cells are
// written.  pout's ghost cells are not written.
template <int NW>
static void abec_gsrb_rb_nbr_nw(const Geometry& g, const AbecCoef& c, MultiFab& pin, MultiFab& pout, const MultiFab& rhs, double omega, bool zero,
                                const DomainBC* bcs, int nbc, int sel, hipStream_t on, bool acc)
{
    const bool walls = !(g.periodic[0] && g.periodic[1] && g.periodic[2]);
    auto& ctx = Context::get();
    hipStream_t strm = on ? on : ctx.stream;
    const Layout& l = *pin.layout;
    const int nbox = l.nlocal();
    const int wpr = l.max_len[0] / 128, rw = NW / wpr;
    const int nty = (l.max_len[1] + (rw - 2) - 1) / (rw - 2);
    // z-chunks: the number of workgroups as close below a whole number of rounds (one 1024-thread workgroup per CU) as the chunk length allows
    const int slots = (int)tune("GSRB_RB_SLOTS", 256), nz = l.max_len[2];
    int best_nch = 1;
    double best_eff = -1.0;
    for (int nch = 1; nch <= std::max(1, nz / 8); ++nch) {
        const int tzc = (nz + nch - 1) / nch;
        if ((nz + tzc - 1) / tzc != nch) continue;
        const long nwg = (long)nbox * nty * nch;
        const long rounds = (nwg + slots - 1) / slots;
        // (every chunk recomputes one plane at each end: the useful fraction of its planes)
        const double eff = (double)nwg / (double)(rounds * slots) * (double)tzc / (double)(tzc + 2);
        if (eff > best_eff + 1.e-9) { best_eff = eff; best_nch = nch; }
    }
    const int tz = (nz + best_nch - 1) / best_nch, nch = (nz + tz - 1) / tz;
    const int ntile = nty * nch, xcd_chunk = tune("GSRB_RB_XCD", 1) != 0 ? (ntile + 7) / 8 : 0, nwg = xcd_chunk > 0 ? 8 * xcd_chunk : ntile;
    const double dhx = c.beta / (g.dx[0] * g.dx[0]), dhy = c.beta / (g.dx[1] * g.dx[1]), dhz = c.beta / (g.dx[2] * g.dx[2]);
    const bool has_a = c.a && c.alpha != 0.0;
    const FabD* At = has_a ? c.a->d_tab : nullptr;
    const FabD* St = c.sig ? c.sig->d_tab : nullptr;
    const FabD Z = pin.h_tab[0];
    const BoxD zb = l.lbox(0);
    const bool xo = l.max_len[0] != g.domain.len(0);         // the boxes are split in x: ghost columns instead of the wrap inside a row
    const long maxface = (long)std::max(l.max_len[0], l.max_len[1]) * std::max(l.max_len[1], l.max_len[2]);
    const dim3 ggrid((unsigned)((maxface + 255) / 256), (unsigned)(6 * nbox));
    const bool rec = !acc && sel == 0 && pin.ncomp == 1 && kernel_probe_begin(PROBE_ABEC_GSRB, (long)l.max_len[0] * l.max_len[1] * l.max_len[2]);
    for (int n = 0; n < pin.ncomp; ++n) {
        BUni bn;
        for (int d = 0; d < 3; ++d) bn.v[d] = c.bu[d] * ((c.tensor_eta && n == d) ? 4.0 / 3.0 : 1.0);
        RbBC rbc;
        for (int d = 0; d < 3; ++d) { rbc.per[d] = 1; rbc.c1lo[d] = rbc.c2lo[d] = rbc.c1hi[d] = rbc.c2hi[d] = 0.0; }
        if (walls) rb_make_bc(g, bcs[n < nbc ? n : 0], rbc);
        for (int d = 0; d < 3; ++d) { rbc.dlo[d] = g.domain.lo[d]; rbc.dhi[d] = g.domain.hi[d]; }
#define IAMRX_RBG(M, HA, WL, SC, SS) if (sel != 1) hipLaunchKernelGGL((k_abec_rb_ghost<M, HA, WL>), ggrid, dim3(256), 0, strm, l.d_boxes, pin.d_tab, rhs.d_tab, At, St, \
                                                       c.alpha, dhx, dhy, dhz, omega, SC, SS, bn, zero ? 1 : 0, n, rbc, xo ? 1 : 0)
#define IAMRX_RBKL(M, HA, WL, XOO, AC, SC, SS) hipLaunchKernelGGL((k_abec_gsrb_rb<M, NW, HA, WL, true, XOO, false, AC>), dim3((unsigned)nwg, (unsigned)nbox), dim3(64 * NW), 0, strm, zb, Z, Z, Z, Z, Z, \
                           c.alpha, dhx, dhy, dhz, omega, SC, SS, bn, wpr, tz, nty, zero ? 1 : 0, n, xcd_chunk, rbc, l.d_boxes, pin.d_tab, pout.d_tab, rhs.d_tab, At, St, sel)
#define IAMRX_RBK(M, HA, WL, XOO, SC, SS) do { if (acc) IAMRX_RBKL(M, HA, WL, XOO, true, SC, SS); else IAMRX_RBKL(M, HA, WL, XOO, false, SC, SS); } while (0)
#define IAMRX_RB(M, HA, WL, SC, SS) IAMRX_RBG(M, HA, WL, SC, SS); if (xo) IAMRX_RBK(M, HA, WL, true, SC, SS); else IAMRX_RBK(M, HA, WL, false, SC, SS)
        if (walls) {
            if (c.sig) { IAMRX_RB(1, false, true, c.sig_comp, c.sig_scale); }
            else if (has_a) { IAMRX_RB(2, true, true, 0, 1.0); }
            else { IAMRX_RB(2, false, true, 0, 1.0); }
        } else {
            if (c.sig) { IAMRX_RB(1, false, false, c.sig_comp, c.sig_scale); }
            else if (has_a) { IAMRX_RB(2, true, false, 0, 1.0); }
            else { IAMRX_RB(2, false, false, 0, 1.0); }
        }
#undef IAMRX_RB
#undef IAMRX_RBK
#undef IAMRX_RBKL
#undef IAMRX_RBG
    }
    if (rec) kernel_probe_end(PROBE_ABEC_GSRB);
}
void abec_gsrb_rb_nbr(const Geometry& g, const AbecCoef& c, MultiFab& pin, MultiFab& pout, const MultiFab& rhs, double omega, bool zero,
                      const DomainBC* bcs, int nbc, int sel, hipStream_t on, bool acc)
{
    IAMRX_ASSERT(abec_gsrb_rb_nbr_ok(g, c, pin, rhs, nbc, bcs) && pin.d_tab != pout.d_tab && pout.ngrow >= 1 && rhs.ncomp == pin.ncomp && pout.ncomp == pin.ncomp);
    if (pin.nlocal() == 0) return;
    // the density form with walls or ghost columns spills at 1024 threads (76 ... 116 bytes of scratch per lane): 12 wavefronts, see abec_gsrb_rb
    const bool walls = !(g.periodic[0] && g.periodic[1] && g.periodic[2]);
    const Layout& l = *pin.layout;
    const int wpr = l.max_len[0] / 128;
    const bool xo = l.max_len[0] != g.domain.len(0);
    if (c.sig && (walls || xo) && 12 % wpr == 0 && 12 / wpr >= 3 && tune("GSRB_RB_NW12", 1) != 0) abec_gsrb_rb_nbr_nw<12>(g, c, pin, pout, rhs, omega, zero, bcs, nbc, sel, on, acc);
    else abec_gsrb_rb_nbr_nw<16>(g, c, pin, pout, rhs, omega, zero, bcs, nbc, sel, on, acc);
}

// IAMRX_ABEC_SIG (1): 0 = the smoother and the residual read the stored face coefficients also where AbecCoef::sig is given
static bool abec_sig_on() { return tune("ABEC_SIG", 1) != 0; }

bool abec_gsrb_zero_ok(const AbecCoef& c, const MultiFab& phi, int nbc, bool wrap, bool has_cf)
{
    return tune("GSRB_ZERO", 1) != 0 && tune("GSRB1_NP", 1) > 0 && wrap && !has_cf && phi.ncomp == 1 && nbc == 1 && c.b[0]->ncomp == 1 && !c.tensor_eta;
}

// the colour passes of this level can apply the domain walls themselves (WallK): one box spanning a domain with non-periodic sides whose
// boundary conditions the two-weight ghost formula covers, and a dispatch of abec_gsrb that ends in k_abec_gsrb / k_abec_gsrb1.  The caller
// then fills periodic ghost cells only (if any) and passes walls_inkernel = true.  Rank-independent.
bool abec_gsrb_walls_inkernel_ok(const Geometry& g, const AbecCoef& c, const MultiFab& phi, int nbc, const DomainBC* bcs, bool cf)
{
    if (tune("GSRB_WALLS_INKERNEL", 1) == 0 || cf || !bcs || nbc < 1 || phi.ncomp > 3 || phi.ngrow < 1) return false;
    const Layout& l = *phi.layout;
    if (l.boxes.size() != 1) return false;
    const BoxD& b = l.boxes[0];
    bool walls = false;
    for (int d = 0; d < 3; ++d) {
        if (b.lo[d] != g.domain.lo[d] || b.hi[d] != g.domain.hi[d] || b.len(d) < 2) return false;
        if (!g.periodic[d]) walls = true;
    }
    if (!walls) return false;
    for (int n = 0; n < phi.ncomp; ++n) { RbBC r; if (!rb_make_bc(g, bcs[n < nbc ? n : 0], r)) return false; }
    // abec_gsrb's dispatch (below): the pair-marching kernels read the ghost cells
    const bool uni = c.b_uniform && c.b[0]->ncomp == 1 && abec_sig_on();
    const int np = (int)tune("GSRB1_NP", 1);
    const int mode = (c.sig && abec_sig_on()) ? 1 : (uni ? 2 : 0);
    const bool pair_ok = mode != 0 && tune("GSRB2", 1) != 0 && phi.ngrow >= 1 && (mode == 2 || c.sig->ngrow >= 1);
    if (np > 0 && phi.ncomp == 1 && nbc == 1 && c.b[0]->ncomp == 1 && !c.tensor_eta) return !pair_ok;
    if (uni && phi.ncomp > 1 && c.b[0]->ncomp == 1 && phi.ngrow >= 1 && tune("GSRB2", 1) != 0 && tune("GSRB2_MULTI", 0) != 0) return false;
    return true;
}

void abec_gsrb(const Geometry& g, const AbecCoef& c, MultiFab& phi, const MultiFab& rhs, int redblack, double omega, const DomainBC* bcs, int nbc, bool shell_only,
               bool wrap, const MultiFab* cfm, const CfTab* cftab, bool cf_maintain_ghosts, bool phi_is_zero, bool walls_inkernel)
{
    // (the wall formulas travel as a pointer to a small device table, null when off: 300 bytes more of kernel arguments cost the colour
    // passes of the small levels 1 us each)
    const WallK* wk = nullptr;
    if (walls_inkernel) {
        IAMRX_ASSERT(!shell_only && !wrap && cfm == nullptr && phi.ncomp <= 3);      // (the caller asked abec_gsrb_walls_inkernel_ok)
        WallK h;
        std::memset(&h, 0, sizeof(h));
        h.on = 1;
        for (int n = 0; n < 3; ++n) {
            RbBC r;
            rb_make_bc(g, bcs[n < nbc ? n : 0], r);
            for (int d = 0; d < 3; ++d) {
                h.per[d] = r.per[d];
                h.c1lo[n][d] = r.c1lo[d]; h.c2lo[n][d] = r.c2lo[d]; h.c1hi[n][d] = r.c1hi[d]; h.c2hi[n][d] = r.c2hi[d];
            }
        }
        // device copies of the few distinct tables of a run, kept for its life
        static std::vector<std::pair<WallK, WallK*>> tabs;
        for (auto& t : tabs) if (std::memcmp(&t.first, &h, sizeof(h)) == 0) wk = t.second;
        if (!wk) {
            WallK* d = nullptr;
            IAMRX_HIP_CHECK(hipMalloc(&d, sizeof(WallK)));
            IAMRX_HIP_CHECK(hipMemcpy(d, &h, sizeof(WallK), hipMemcpyHostToDevice));
            tabs.emplace_back(h, d);
            wk = d;
        }
    }
    IAMRX_ASSERT(!phi_is_zero || (!shell_only && abec_gsrb_zero_ok(c, phi, nbc, wrap || walls_inkernel, cfm != nullptr)));
    const int zero = phi_is_zero ? 1 : 0;
    CfC1 cfc;
    cfc.maxorder = cftab ? cftab->maxorder : 2;
    cfc.maintain = (cf_maintain_ghosts && cfm && cftab && phi.ncomp == 1 && !shell_only) ? 1 : 0;
    for (int d = 0; d < 3; ++d) for (int q = 0; q < 3; ++q) {
        cfc.c1[d][q] = cftab ? cftab->c[d][q][1] : 0.0; cfc.c2[d][q] = cftab ? cftab->c[d][q][2] : 0.0; cfc.c3[d][q] = cftab ? cftab->c[d][q][3] : 0.0;
    }
    const FabD* cft = (cfm && cftab) ? cfm->d_tab : nullptr;
    if (phi.nlocal() == 0) return;
    auto& ctx = Context::get();
    const Layout& l = *phi.layout;
    Tiling t = pair_tiling(l, 8);                 // (a flat tile list on a level of unequal boxes: launch.h)
    const double dhx = c.beta / (g.dx[0] * g.dx[0]), dhy = c.beta / (g.dx[1] * g.dx[1]), dhz = c.beta / (g.dx[2] * g.dx[2]);
    GsrbBC gb = make_gsrb_bc(g, bcs, nbc);
    // the scalar (MAC projection / scalar diffusion) colour pass over whole boxes can be timed in place (bench.py)
    const bool rec = phi.ncomp == 1 && !shell_only && kernel_probe_begin(PROBE_ABEC_GSRB, (long)l.max_len[0] * l.max_len[1] * l.max_len[2]);
    BUni bu;
    for (int d = 0; d < 3; ++d) bu.v[d] = c.bu[d];
    const bool uni = c.b_uniform && c.b[0]->ncomp == 1 && abec_sig_on();
    // IAMRX_GSRB1_NP (1): planes in flight per thread of the pipelined one-component pass (0: the general kernel)
    const int np = (int)tune("GSRB1_NP", 1);
    const int mode = (c.sig && abec_sig_on()) ? 1 : (uni ? 2 : 0);
    const bool pair_ok = mode != 0 && tune("GSRB2", 1) != 0 && phi.ngrow >= 1 && (mode == 2 || c.sig->ngrow >= 1);
    // with coarse/fine faces the pipelined form would read a ghost cell that cf_maintain rewrote for the plane before: one plane in flight
    if (np > 0 && phi.ncomp == 1 && !shell_only && nbc == 1 && c.b[0]->ncomp == 1 && !c.tensor_eta) {
        const FabD *t0 = mode == 1 ? c.sig->d_tab : c.b[0]->d_tab, *t1 = mode == 1 ? c.sig->d_tab : c.b[1]->d_tab, *t2 = mode == 1 ? c.sig->d_tab : c.b[2]->d_tab;
#define IAMRX_GS1(M, N) if (cft) IAMRX_GS1C(M, 1, true); else IAMRX_GS1C(M, N, false)
#define IAMRX_GS1C(M, N, C) hipLaunchKernelGGL((k_abec_gsrb1<M, N, C>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, phi.d_tab, rhs.d_tab, \
                           c.a ? c.a->d_tab : nullptr, t0, t1, t2, c.alpha, dhx, dhy, dhz, redblack, omega, gb, wrap ? 1 : 0, c.sig_comp, c.sig_scale, bu, cft, cfc, zero, wk)
        // IAMRX_GSRB2 (1): the pair-marching kernel where the coefficients are not arrays (needs a ghost layer for its 16-byte loads)
        if (pair_ok) {
            Tiling t2 = pair_tiling(l, (int)tune("GSRB2_TZ", 32));
#define IAMRX_GS2(M, C, MT, SG, SC, SS) hipLaunchKernelGGL((k_abec_gsrb2<M, C, MT>), t2.grid(), Tiling::block(), 0, ctx.stream, t2, l.d_boxes, phi.d_tab, rhs.d_tab, \
                                   c.a ? c.a->d_tab : nullptr, SG, c.alpha, dhx, dhy, dhz, redblack, omega, gb, wrap ? 1 : 0, SC, SS, bu, cft, cfc, zero, 0, 0)
            const bool mt = cft && cfc.maintain;
            // one box strictly inside the domain (no neighbour box, no domain face, no periodic image next to it): all of its ghost cells are
            // coarse/fine ghost cells and the maintaining colour pass takes the masks as constants.  IAMRX_GSRB2_ALLCF = 0: loads them.
            bool allcf = mt && l.boxes.size() == 1 && l.nlocal() == 1 && tune("GSRB2_ALLCF", 1) != 0;
            for (int d = 0; d < 3 && allcf; ++d) allcf = l.boxes[0].lo[d] > g.domain.lo[d] && l.boxes[0].hi[d] < g.domain.hi[d];
            if (allcf) {
                if (mode == 1) hipLaunchKernelGGL((k_abec_gsrb2<1, true, true, true>), t2.grid(), Tiling::block(), 0, ctx.stream, t2, l.d_boxes, phi.d_tab, rhs.d_tab,
                                   c.a ? c.a->d_tab : nullptr, c.sig->d_tab, c.alpha, dhx, dhy, dhz, redblack, omega, gb, wrap ? 1 : 0, c.sig_comp, c.sig_scale, bu, cft, cfc, zero, 0, 0);
                else hipLaunchKernelGGL((k_abec_gsrb2<2, true, true, true>), t2.grid(), Tiling::block(), 0, ctx.stream, t2, l.d_boxes, phi.d_tab, rhs.d_tab,
                                   c.a ? c.a->d_tab : nullptr, nullptr, c.alpha, dhx, dhy, dhz, redblack, omega, gb, wrap ? 1 : 0, 0, 1.0, bu, cft, cfc, zero, 0, 0);
            }
            else if (mode == 1) {
                if (mt) IAMRX_GS2(1, true, true, c.sig->d_tab, c.sig_comp, c.sig_scale);
                else if (cft) IAMRX_GS2(1, true, false, c.sig->d_tab, c.sig_comp, c.sig_scale);
                else IAMRX_GS2(1, false, false, c.sig->d_tab, c.sig_comp, c.sig_scale);
            } else {
                if (mt) IAMRX_GS2(2, true, true, nullptr, 0, 1.0);
                else if (cft) IAMRX_GS2(2, true, false, nullptr, 0, 1.0);
                else IAMRX_GS2(2, false, false, nullptr, 0, 1.0);
            }
#undef IAMRX_GS2
        }
        else if (np >= 4) { if (mode == 1) { IAMRX_GS1(1, 4); } else if (mode == 2) { IAMRX_GS1(2, 4); } else { IAMRX_GS1(0, 4); } }
        else if (np >= 2) { if (mode == 1) { IAMRX_GS1(1, 2); } else if (mode == 2) { IAMRX_GS1(2, 2); } else { IAMRX_GS1(0, 2); } }
        else { if (mode == 1) { IAMRX_GS1(1, 1); } else if (mode == 2) { IAMRX_GS1(2, 1); } else { IAMRX_GS1(0, 1); } }
#undef IAMRX_GS1
#undef IAMRX_GS1C
    }
    else if (phi.ncomp == 1 && c.sig && !c.tensor_eta && abec_sig_on())
        hipLaunchKernelGGL((k_abec_gsrb<false, 1>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, phi.d_tab, rhs.d_tab,
                           c.a ? c.a->d_tab : nullptr, c.sig->d_tab, c.sig->d_tab, c.sig->d_tab,
                           c.alpha, dhx, dhy, dhz, redblack, omega, 1, 1, gb, shell_only ? 1 : 0, 0, wrap ? 1 : 0, cft, cfc, c.sig_comp, c.sig_scale, BUni(), wk);
    else if (uni && phi.ncomp > 1 && !shell_only && c.b[0]->ncomp == 1 && phi.ngrow >= 1 && tune("GSRB2", 1) != 0 && tune("GSRB2_MULTI", 0) != 0) {
        // IAMRX_GSRB2_MULTI = 1 (off by default: measured equal, 3 x 112 us against 336 us at 256^3 -- both forms move the half-used lines of
        // the red-black layout at the achievable HBM rate): several components with constant coefficients (the tensor solves of a
        // constant-viscosity run, the unit-coefficient tensor solve of diffuse_tensor_Vsync) as one pair-marching launch per component
        // with its own constants b_d (x 4/3 on the normal component of the tensor operator) -- the expressions of k_abec_gsrb<true, 2>
        Tiling t2 = pair_tiling(l, (int)tune("GSRB2_TZ", 32));
        for (int n = 0; n < phi.ncomp; ++n) {
            BUni bn;
            for (int d = 0; d < 3; ++d) bn.v[d] = bu.v[d] * ((c.tensor_eta && n == d) ? 4.0 / 3.0 : 1.0);
            const int bq = gb.nbc == 1 ? 0 : (n < 3 ? n : 0);
#define IAMRX_GS2M(C) hipLaunchKernelGGL((k_abec_gsrb2<2, C, false>), t2.grid(), Tiling::block(), 0, ctx.stream, t2, l.d_boxes, phi.d_tab, rhs.d_tab, \
                                  c.a ? c.a->d_tab : nullptr, nullptr, c.alpha, dhx, dhy, dhz, redblack, omega, gb, wrap ? 1 : 0, 0, 1.0, bn, cft, cfc, 0, n, bq)
            if (cft) IAMRX_GS2M(true); else IAMRX_GS2M(false);
#undef IAMRX_GS2M
        }
    }
    else if (uni && phi.ncomp > 1)
        hipLaunchKernelGGL((k_abec_gsrb<true, 2>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, phi.d_tab, rhs.d_tab,
                           c.a ? c.a->d_tab : nullptr, c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab,
                           c.alpha, dhx, dhy, dhz, redblack, omega, phi.ncomp, 1, gb, shell_only ? 1 : 0, c.tensor_eta, wrap ? 1 : 0, cft, cfc, 0, 1.0, bu, wk);
    else if (uni)
        hipLaunchKernelGGL((k_abec_gsrb<false, 2>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, phi.d_tab, rhs.d_tab,
                           c.a ? c.a->d_tab : nullptr, c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab,
                           c.alpha, dhx, dhy, dhz, redblack, omega, phi.ncomp, 1, gb, shell_only ? 1 : 0, c.tensor_eta, wrap ? 1 : 0, cft, cfc, 0, 1.0, bu, wk);
    else if (phi.ncomp > 1 && c.b[0]->ncomp == 1)
        hipLaunchKernelGGL((k_abec_gsrb<true, 0>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, phi.d_tab, rhs.d_tab,
                           c.a ? c.a->d_tab : nullptr, c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab,
                           c.alpha, dhx, dhy, dhz, redblack, omega, phi.ncomp, c.b[0]->ncomp, gb, shell_only ? 1 : 0, c.tensor_eta, wrap ? 1 : 0, cft, cfc, 0, 1.0, BUni(), wk);
    else
        hipLaunchKernelGGL((k_abec_gsrb<false, 0>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, phi.d_tab, rhs.d_tab,
                           c.a ? c.a->d_tab : nullptr, c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab,
                           c.alpha, dhx, dhy, dhz, redblack, omega, phi.ncomp, c.b[0]->ncomp, gb, shell_only ? 1 : 0, c.tensor_eta, wrap ? 1 : 0, cft, cfc, 0, 1.0, BUni(), wk);
    if (rec) kernel_probe_end(PROBE_ABEC_GSRB);
}

// ---------------------------------------------------------------------------- fused red+black sweep
// One pass over the level does the red update of every cell and the black update of every cell that is not on the
// surface of its box; the black cells of that one-cell shell need ghost values that depend on the red update of other
// boxes / the domain BC and are done afterwards by k_abec_gsrb(shell_only) once the ghosts are refreshed.  A workgroup
// owns a TXxTY column and marches in z: phi planes live in a rolling LDS window (k-1..k+2), the red update of plane k+1
// runs one plane ahead of the black update of plane k, and the red values of the one-cell ring around the column are
// recomputed instead of exchanged.  Out of place (pin -> pout): other workgroups read pin in their rings.
// Arithmetic per cell is that of k_abec_gsrb, so the result equals applyBC+red, applyBC+black bit for bit.
template <int TX, int TY>
__global__ void __launch_bounds__(256) k_abec_gsrb_fused(const BoxD* __restrict__ boxes, const FabD* __restrict__ pint, const FabD* __restrict__ poutt,
    const FabD* __restrict__ rhst, const FabD* __restrict__ at, const FabD* __restrict__ bxt, const FabD* __restrict__ byt, const FabD* __restrict__ bzt,
    double alpha, double dhx, double dhy, double dhz, double omega, int ncomp, int bnc, GsrbBC bc, int ntx, int nty, int kc, int tens)
{
    static_assert(TX * TY == 512, "one red and one black cell per thread and plane");
    constexpr int FX = TX + 4, FY = TY + 4, NLD = (FX * FY + 255) / 256;
    constexpr int NRING = (TX + 2) + (TX + 2) + TY + TY;          // cells of the one-cell ring around the column
    __shared__ double P[4][FY][FX];
    const int fab = blockIdx.y;
    const BoxD b = boxes[fab];
    const int bid = blockIdx.x;
    const int tix = bid % ntx, r1 = bid / ntx, tiy = r1 % nty, ck = r1 / nty;
    const int tx0 = b.lo[0] + tix * TX, ty0 = b.lo[1] + tiy * TY, k0 = b.lo[2] + ck * kc;
    if (tx0 > b.hi[0] || ty0 > b.hi[1] || k0 > b.hi[2]) return;
    const int txe = min(tx0 + TX - 1, b.hi[0]), tye = min(ty0 + TY - 1, b.hi[1]), k1 = min(k0 + kc - 1, b.hi[2]);
    const FabD pin = pint[fab], pout = poutt[fab], rhs = rhst[fab], bX = bxt[fab], bY = byt[fab], bZ = bzt[fab];
    const bool has_a = (at != nullptr) && alpha != 0.0;
    FabD A; if (has_a) A = at[fab];
    const int ox = tx0 - 2, oy = ty0 - 2;
    const int tid = threadIdx.x;
    long poff[NLD];
    int lrow[NLD], lcol[NLD];
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int idx = min(tid + it * 256, FX * FY - 1);
        lcol[it] = idx % FX; lrow[it] = idx / FX;
        const int gi = min(max(ox + lcol[it], pin.lo[0]), pin.lo[0] + pin.n[0] - 1), gj = min(max(oy + lrow[it], pin.lo[1]), pin.lo[1] + pin.n[1] - 1);
        poff[it] = pin.off(gi, gj, pin.lo[2]);
    }
    const long ppl = (long)pin.n[0] * pin.n[1];
    const int pklo = pin.lo[2], pkhi = pin.lo[2] + pin.n[2] - 1;
    // this thread's pair of cells (x = px, px+1 in row py) and its ring cell (threads < NRING)
    const int px = tx0 + 2 * (tid % (TX / 2)), py = ty0 + tid / (TX / 2);
    int rgx, rgy;
    {
        const int q = tid;
        if (q < TX + 2) { rgx = tx0 - 1 + q; rgy = ty0 - 1; }
        else if (q < 2 * (TX + 2)) { rgx = tx0 - 1 + (q - (TX + 2)); rgy = ty0 + TY; }
        else if (q < 2 * (TX + 2) + TY) { rgx = tx0 - 1; rgy = ty0 + (q - 2 * (TX + 2)); }
        else { rgx = tx0 + TX; rgy = ty0 + (q - 2 * (TX + 2) - TY); }
    }
    // the ring of a clipped column (box narrower than the tile) hugs the clipped column
    const bool ring_thread = tid < NRING;
    struct Cf { double bxm, bxp, bym, byp, bzm, bzp, aa, r; };
    for (int n = 0; n < ncomp; ++n) {
        const int nb = bnc == 1 ? 0 : n;
        const int nq = bc.nbc == 1 ? 0 : (n < 3 ? n : 0);
        const double sx = (tens && n == 0) ? 4.0 / 3.0 : 1.0, sy = (tens && n == 1) ? 4.0 / 3.0 : 1.0, sz = (tens && n == 2) ? 4.0 / 3.0 : 1.0;
        const FabD::gdouble* pp = (const FabD::gdouble*)pin.p + pin.cs * n;
        auto fetch_plane = [&](int m, double (&v)[NLD]) {
            const int mc = min(max(m, pklo), pkhi);
#pragma unroll
            for (int it = 0; it < NLD; ++it) v[it] = pp[poff[it] + ppl * (mc - pklo)];
        };
        auto store_plane = [&](int m, const double (&v)[NLD]) {
#pragma unroll
            for (int it = 0; it < NLD; ++it) if (tid + it * 256 < FX * FY) P[m & 3][lrow[it]][lcol[it]] = v[it];
        };
        // coefficients of cell (i,j,m); indices clamped into the box so that the loads are unconditional
        auto fetch_cf = [&](int i, int j, int m) {
            const int ic = min(max(i, b.lo[0]), b.hi[0]), jc = min(max(j, b.lo[1]), b.hi[1]), mc = min(max(m, b.lo[2]), b.hi[2]);
            Cf c;
            c.bxm = bX(ic, jc, mc, nb) * sx; c.bxp = bX(ic + 1, jc, mc, nb) * sx;
            c.bym = bY(ic, jc, mc, nb) * sy; c.byp = bY(ic, jc + 1, mc, nb) * sy;
            c.bzm = bZ(ic, jc, mc, nb) * sz; c.bzp = bZ(ic, jc, mc + 1, nb) * sz;
            c.aa = has_a ? A(ic, jc, mc, 0) : 0.0;
            c.r = rhs(ic, jc, mc, n);
            return c;
        };
        auto update = [&](int i, int j, int m, const Cf& c) {
            const int lx = i - ox, ly = j - oy;
            const double cf0 = (i == bc.dlo[0]) ? bc.cflo[nq][0] : 0.0, cf3 = (i == bc.dhi[0]) ? bc.cfhi[nq][0] : 0.0;
            const double cf1 = (j == bc.dlo[1]) ? bc.cflo[nq][1] : 0.0, cf4 = (j == bc.dhi[1]) ? bc.cfhi[nq][1] : 0.0;
            const double cf2 = (m == bc.dlo[2]) ? bc.cflo[nq][2] : 0.0, cf5 = (m == bc.dhi[2]) ? bc.cfhi[nq][2] : 0.0;
            const double aa = has_a ? alpha * c.aa : 0.0;
            const double gamma = aa + dhx * (c.bxm + c.bxp) + dhy * (c.bym + c.byp) + dhz * (c.bzm + c.bzp);
            const double g_m_d = gamma - (dhx * (c.bxm * cf0 + c.bxp * cf3) + dhy * (c.bym * cf1 + c.byp * cf4) + dhz * (c.bzm * cf2 + c.bzp * cf5));
            const double (*Pc)[FX] = P[m & 3];
            const double rho = dhx * (c.bxm * Pc[ly][lx - 1] + c.bxp * Pc[ly][lx + 1])
                             + dhy * (c.bym * Pc[ly - 1][lx] + c.byp * Pc[ly + 1][lx])
                             + dhz * (c.bzm * P[(m - 1) & 3][ly][lx] + c.bzp * P[(m + 1) & 3][ly][lx]);
            const double p0 = Pc[ly][lx];
            const double res = c.r - (gamma * p0 - rho);
            return p0 + omega / g_m_d * res;
        };
        auto in_box = [&](int i, int j, int m) { return i >= b.lo[0] && i <= b.hi[0] && j >= b.lo[1] && j <= b.hi[1] && m >= b.lo[2] && m <= b.hi[2]; };
        // red update of plane m: the thread's own red cell and, for ring threads, a red ring cell
        auto red_x = [&](int m) { return px + ((px + py + m) & 1); };
        auto red_plane = [&](int m, const Cf& own, const Cf& rg) {
            if (m < b.lo[2] || m > b.hi[2]) return;
            const int i = red_x(m);
            if (i <= txe && py <= tye) P[m & 3][py - oy][i - ox] = update(i, py, m, own);
            if (ring_thread && ((rgx + rgy + m) & 1) == 0 && in_box(rgx, rgy, m) && rgx <= txe + 1 && rgy <= tye + 1)
                P[m & 3][rgy - oy][rgx - ox] = update(rgx, rgy, m, rg);
        };
        double pv[NLD];
        fetch_plane(k0 - 2, pv); store_plane(k0 - 2, pv);
        fetch_plane(k0 - 1, pv); store_plane(k0 - 1, pv);
        fetch_plane(k0, pv); store_plane(k0, pv);
        fetch_plane(k0 + 1, pv); store_plane(k0 + 1, pv);
        {
            const Cf o1 = fetch_cf(red_x(k0 - 1), py, k0 - 1), g1 = fetch_cf(rgx, rgy, k0 - 1);
            const Cf o2 = fetch_cf(red_x(k0), py, k0), g2 = fetch_cf(rgx, rgy, k0);
            __syncthreads();
            red_plane(k0 - 1, o1, g1);
            __syncthreads();
            red_plane(k0, o2, g2);
        }
        __syncthreads();
        for (int k = k0; k <= k1; ++k) {
            // everything this step needs from memory is requested up front
            fetch_plane(k + 2, pv);
            const int ir = red_x(k + 1);                  // red cell of plane k+1
            const int ib = px + 1 - ((px + py + k) & 1);  // black cell of plane k
            const Cf cr = fetch_cf(ir, py, k + 1), cg = fetch_cf(rgx, rgy, k + 1), cb = fetch_cf(ib, py, k);
            store_plane(k + 2, pv);                       // slot of plane k-2, no longer needed
            __syncthreads();
            red_plane(k + 1, cr, cg);
            __syncthreads();
            // output plane k: the red cell is final; the black cell gets its update unless it lies on the box surface
            if (py <= tye) {
                const int irk = red_x(k);
                if (irk <= txe) pout(irk, py, k, n) = P[k & 3][py - oy][irk - ox];
                if (ib <= txe) {
                    const bool inner = ib > b.lo[0] && ib < b.hi[0] && py > b.lo[1] && py < b.hi[1] && k > b.lo[2] && k < b.hi[2];
                    pout(ib, py, k, n) = inner ? update(ib, py, k, cb) : P[k & 3][py - oy][ib - ox];
                }
            }
            __syncthreads();
        }
    }
}

// one full red+black sweep, phi_in (ghost cells filled by the caller) -> phi_out (valid cells; the black cells on box surfaces still
// hold their old value: refresh the ghost cells of phi_out, then call abec_gsrb(..., redblack = 1, shell_only = true))
void abec_gsrb_fused(const Geometry& g, const AbecCoef& c, const MultiFab& phi_in, MultiFab& phi_out, const MultiFab& rhs, double omega,
                     const DomainBC* bcs, int nbc)
{
    if (phi_in.nlocal() == 0) return;
    IAMRX_ASSERT(phi_in.d_tab != phi_out.d_tab && phi_in.ngrow >= 1);
    auto& ctx = Context::get();
    const Layout& l = *phi_in.layout;
    constexpr int TX = 64, TY = 8;
    const int ntx = (l.max_len[0] + TX - 1) / TX, nty = (l.max_len[1] + TY - 1) / TY;
    const int nk = l.max_len[2];
    int kc = nk;
    while (kc > 16 && (long)ntx * nty * ((nk + kc - 1) / kc) * l.nlocal() < 1536) kc = (kc + 1) / 2;
    const int nck = (nk + kc - 1) / kc;
    const double dhx = c.beta / (g.dx[0] * g.dx[0]), dhy = c.beta / (g.dx[1] * g.dx[1]), dhz = c.beta / (g.dx[2] * g.dx[2]);
    GsrbBC gb = make_gsrb_bc(g, bcs, nbc);
    dim3 grid((unsigned)(ntx * nty * nck), (unsigned)l.nlocal());
    hipLaunchKernelGGL((k_abec_gsrb_fused<TX, TY>), grid, dim3(256), 0, ctx.stream, l.d_boxes, phi_in.d_tab, phi_out.d_tab, rhs.d_tab,
                       c.a ? c.a->d_tab : nullptr, c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab,
                       c.alpha, dhx, dhy, dhz, omega, phi_in.ncomp, c.b[0]->ncomp, gb, ntx, nty, kc, c.tensor_eta);
}

// ---------------------------------------------------------------------------- residual / apply
// max norm of what a residual launch wrote, as a by-product: wave maximum -> one atomicMax per wavefront on the bit pattern of the
// (non-negative, NaN -> +inf) value.  Order independent, hence deterministic.
__device__ __forceinline__ void norm_commit(double mx, unsigned long long* out)
{
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.0) {
        // the running maximum only grows: a plain read filters out almost every wavefront before the atomic (262 k atomics on one address
        // cost 130 us per 256^3 launch without it)
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
        if (bits > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, bits);
    }
}
__device__ __forceinline__ double norm_term(double v) { const double a = fabs(v); return a == a ? a : INFINITY; }

template <int BMODE>
__global__ void __launch_bounds__(256) k_abec_residual(Tiling t, const BoxD* __restrict__ boxes,
    const FabD* __restrict__ outt, const FabD* __restrict__ phit, const FabD* __restrict__ rhst, const FabD* __restrict__ at,
    const FabD* __restrict__ bxt, const FabD* __restrict__ byt, const FabD* __restrict__ bzt,
    double alpha, double dhx, double dhy, double dhz, int ncomp, int bnc, int tens, unsigned long long* __restrict__ normout,
    int sig_comp, double sig_scale, BUni bu)
{
    constexpr bool SIG = BMODE == 1, UNI = BMODE == 2;
    const int fab = tile_fab(t);
    const BoxD b = boxes[fab];
    int i, j, k0, k1;
    double mx = 0.0;
    if (!tile_ijk(t, b, i, j, k0, k1)) { if (normout) norm_commit(mx, normout); return; }
    const FabD out = outt[fab], phi = phit[fab], bX = bxt[fab], bY = byt[fab], bZ = bzt[fab];
    const bool has_rhs = rhst != nullptr;
    FabD rhs; if (has_rhs) rhs = rhst[fab];
    const bool has_a = (at != nullptr) && alpha != 0.0;
    FabD A; if (has_a) A = at[fab];
    for (int n = 0; n < ncomp; ++n) {
        const int nb = bnc == 1 ? 0 : n;
        const double sx = (tens && n == 0) ? 4.0 / 3.0 : 1.0, sy = (tens && n == 1) ? 4.0 / 3.0 : 1.0, sz = (tens && n == 2) ? 4.0 / 3.0 : 1.0;
        double pm = phi(i, j, k0 - 1, n), p0 = phi(i, j, k0, n);
        // SIG: the cell-centred values march with the planes; the z-face coefficient of plane k + 1 is the upper one of plane k
        double sm = 0.0, s0 = 0.0, bzlo = 0.0;
        if (SIG) { sm = bX(i, j, k0 - 1, sig_comp); s0 = bX(i, j, k0, sig_comp); bzlo = sig_scale / (0.5 * (sm + s0)); }
        for (int k = k0; k <= k1; ++k) {
            const double pp = phi(i, j, k + 1, n);
            const double ax = has_a ? alpha * A(i, j, k, 0) * p0 : 0.0;
            double bxm, bxp, bym, byp, bzm, bzp;
            if (SIG) {                         // mac_bcoef's expression, lower cell first
                const double sp = bX(i, j, k + 1, sig_comp);
                bxm = sig_scale / (0.5 * (bX(i - 1, j, k, sig_comp) + s0)); bxp = sig_scale / (0.5 * (s0 + bX(i + 1, j, k, sig_comp)));
                bym = sig_scale / (0.5 * (bX(i, j - 1, k, sig_comp) + s0)); byp = sig_scale / (0.5 * (s0 + bX(i, j + 1, k, sig_comp)));
                bzm = bzlo; bzp = sig_scale / (0.5 * (s0 + sp));
                bzlo = bzp; sm = s0; s0 = sp;
            } else if (UNI) {
                bxm = bxp = bu.v[0] * sx; bym = byp = bu.v[1] * sy; bzm = bzp = bu.v[2] * sz;
            } else {
                bxm = bX(i, j, k, nb) * sx; bxp = bX(i + 1, j, k, nb) * sx;
                bym = bY(i, j, k, nb) * sy; byp = bY(i, j + 1, k, nb) * sy;
                bzm = bZ(i, j, k, nb) * sz; bzp = bZ(i, j, k + 1, nb) * sz;
            }
            const double y = ax
                - dhx * (bxp * (phi(i + 1, j, k, n) - p0) - bxm * (p0 - phi(i - 1, j, k, n)))
                - dhy * (byp * (phi(i, j + 1, k, n) - p0) - bym * (p0 - phi(i, j - 1, k, n)))
                - dhz * (bzp * (pp - p0) - bzm * (p0 - pm));
            const double o = has_rhs ? rhs(i, j, k, n) - y : y;
            out(i, j, k, n) = o;
            mx = fmax(mx, norm_term(o));
            pm = p0; p0 = pp;
        }
    }
    if (normout) norm_commit(mx, normout);
}


// norm_out != null: *norm_out = max norm of `out` over all components and ranks, computed by the launch that writes it last (saves
// the separate norm pass of every multigrid iteration)
static bool abec_residual_pairs(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& phi, const MultiFab& rhs, unsigned long long* d_norm);

void abec_residual(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& phi, const MultiFab* rhs, double* norm_out)
{
    auto& ctx = Context::get();
    static unsigned long long* d_norm = nullptr;
    if (norm_out && !d_norm) IAMRX_HIP_CHECK(hipMalloc(&d_norm, 2 * sizeof(unsigned long long)));
    if (phi.nlocal() > 0) {
        const Layout& l = *phi.layout;
        Tiling t = level_tiling(l, cell_type(), 0, 8, true);        // (a flat tile list on a level of unequal boxes: launch.h)
        const double dhx = c.beta / (g.dx[0] * g.dx[0]), dhy = c.beta / (g.dx[1] * g.dx[1]), dhz = c.beta / (g.dx[2] * g.dx[2]);
        if (norm_out) IAMRX_HIP_CHECK(hipMemsetAsync(d_norm, 0, sizeof(unsigned long long), ctx.stream));
        if (rhs && abec_residual_pairs(g, c, out, phi, *rhs, norm_out ? d_norm : nullptr)) { /* done by the pair march */ }
        else {
        BUni bu;
        for (int d = 0; d < 3; ++d) bu.v[d] = c.bu[d];
        if (phi.ncomp == 1 && c.sig && !c.tensor && !c.tensor_eta && abec_sig_on())
            hipLaunchKernelGGL(k_abec_residual<1>, t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, out.d_tab, phi.d_tab,
                               rhs ? rhs->d_tab : nullptr, c.a ? c.a->d_tab : nullptr, c.sig->d_tab, c.sig->d_tab, c.sig->d_tab,
                               c.alpha, dhx, dhy, dhz, 1, 1, 0, norm_out ? d_norm : nullptr, c.sig_comp, c.sig_scale, bu);
        else if (c.b_uniform && c.b[0]->ncomp == 1 && abec_sig_on())
            hipLaunchKernelGGL(k_abec_residual<2>, t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, out.d_tab, phi.d_tab,
                               rhs ? rhs->d_tab : nullptr, c.a ? c.a->d_tab : nullptr, c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab,
                               c.alpha, dhx, dhy, dhz, phi.ncomp, 1, c.tensor_eta, (norm_out && !c.tensor) ? d_norm : nullptr, 0, 1.0, bu);
        else
        hipLaunchKernelGGL(k_abec_residual<0>, t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, out.d_tab, phi.d_tab,
                           rhs ? rhs->d_tab : nullptr, c.a ? c.a->d_tab : nullptr, c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab,
                           c.alpha, dhx, dhy, dhz, phi.ncomp, c.b[0]->ncomp, c.tensor_eta, (norm_out && !c.tensor) ? d_norm : nullptr, 0, 1.0, bu);
        }
        if (c.tensor) tensor_cross_terms_sub(g, c, out, phi, rhs ? -1.0 : 1.0, norm_out ? d_norm : nullptr);
    }
    if (norm_out) {
        // the bit pattern of a non-negative double orders like the double: the ranks' maxima are combined in place on the device
        // (Comm::allreduce_device), then read back once
        double v = 0.0;
        const bool global = !phi.layout->replicated && ctx.comm->nranks > 1;
        if (phi.nlocal() == 0 && global) IAMRX_HIP_CHECK(hipMemsetAsync(d_norm, 0, sizeof(unsigned long long), ctx.stream));
        if (global) ctx.comm->allreduce_device(reinterpret_cast<double*>(d_norm), 1, ReduceOp::Max, ctx.stream);
        if (phi.nlocal() > 0 || global) {
            unsigned long long bits = 0;
            IAMRX_HIP_CHECK(hipMemcpyAsync(&bits, d_norm, sizeof(bits), hipMemcpyDeviceToHost, ctx.stream));
            ctx.sync();
            std::memcpy(&v, &bits, sizeof(v));
        }
        *norm_out = v;
    }
}

// ---------------------------------------------------------------------------- residual + restriction in one pass
// The down-leg of a V-cycle needs the residual of the smoothed correction only as the coarse level's right-hand side.  A thread owns a
// coarse cell column (I, J) and marches through the fine planes: per plane it forms the four residuals of its fine cells
// (2I, 2I+1) x (2J, 2J+1) -- k_abec_residual's expression, coefficients from the cell-centred array (BMODE 1) or constants (BMODE 2) --
// from row pairs read with 16-byte loads and kept for three planes, x-neighbours from the adjacent lane, and adds them in k_cc_restrict's
// order; every second plane it writes 0.125 x the sum.  The fine residual (a full array written and read again) never exists.  Same doubles
// as abec_residual followed by cc_restrict.  phi: ghost cells filled; no 'a' term.
// RESTRICT = false: the same march writes the four fine residuals per plane instead (ct: the fine output array) and reduces their max norm
// (normout, see norm_commit) -- the residual of the convergence test with 7 sixteen-byte loads per two cells instead of 22 eight-byte ones.
template <int BMODE, bool RESTRICT>
__global__ void __launch_bounds__(256) k_abec_resid_restrict(Tiling t, const BoxD* __restrict__ cboxes, const FabD* __restrict__ ct,
    const FabD* __restrict__ phit, const FabD* __restrict__ rhst, const FabD* __restrict__ sgt,
    double dhx, double dhy, double dhz, int sig_comp, double sig_scale, BUni bu, unsigned long long* __restrict__ normout, int wrap)
{
    const int fab = blockIdx.y;
    const BoxD cb = cboxes[fab];
    int I, J, K0, K1;
    double mx = 0.0;
    if (!tile_ijk(t, cb, I, J, K0, K1)) { if (!RESTRICT && normout) norm_commit(mx, normout); return; }
    // wrap: one box spanning a periodic domain -- phi's periodic images are read from its valid cells (no ghost fill in front of the launch)
    const int flo0 = 2 * cb.lo[0], fhi0 = 2 * cb.hi[0] + 1, flo1 = 2 * cb.lo[1], fhi1 = 2 * cb.hi[1] + 1, flo2 = 2 * cb.lo[2], fhi2 = 2 * cb.hi[2] + 1;
    auto wj = [&](int j) { return wrap ? (j < flo1 ? fhi1 : (j > fhi1 ? flo1 : j)) : j; };
    auto wk = [&](int k) { return wrap ? (k < flo2 ? fhi2 : (k > fhi2 ? flo2 : k)) : k; };
    const FabD crse = ct[fab], phi = phit[fab], rhs = rhst[fab];
    FabD S; if (BMODE == 1) S = sgt[fab];
    const int bx = 1 << t.bxs, tx = (int)threadIdx.x & (bx - 1);
    const bool laneL = tx > 0 && I > cb.lo[0], laneR = tx < bx - 1 && I < cb.hi[0];
    const int iL = 2 * I, iR = iL + 1, j0 = 2 * J;
    auto bco = [&](double a, double b) { return sig_scale / (0.5 * (a + b)); };       // mac_bcoef's expression, lower cell first
    // rows j0, j0 + 1 at planes k - 1 (b), k (c), k + 1 (a)
    D2 pb[2], pc[2], pa[2], sb[2], sc[2], sa[2];
    int k = 2 * K0;
    for (int r = 0; r < 2; ++r) {
        pb[r] = ld2(phi, iL, j0 + r, wk(k - 1), 0); pc[r] = ld2(phi, iL, j0 + r, k, 0);
        if (BMODE == 1) { sb[r] = ld2(S, iL, j0 + r, k - 1, sig_comp); sc[r] = ld2(S, iL, j0 + r, k, sig_comp); }
    }
    for (int K = K0; K <= K1; ++K) {
        double s = 0.0;
        for (int kr = 0; kr < 2; ++kr, ++k) {
            for (int r = 0; r < 2; ++r) {
                pa[r] = ld2(phi, iL, j0 + r, wk(k + 1), 0);
                if (BMODE == 1) sa[r] = ld2(S, iL, j0 + r, k + 1, sig_comp);
            }
            const D2 pS = ld2(phi, iL, wj(j0 - 1), k, 0), pN = ld2(phi, iL, wj(j0 + 2), k, 0);       // the rows below / above the pair of rows
            D2 sS, sN;
            if (BMODE == 1) { sS = ld2(S, iL, j0 - 1, k, sig_comp); sN = ld2(S, iL, j0 + 2, k, sig_comp); }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int j = j0 + r;
                const D2 rr = ld2(rhs, iL, j, k, 0);
                // x-neighbours of the pair: the adjacent lanes' near cells (or loads at the ends of the row segment)
                const double fl = __shfl_up(pc[r].r, 1, 64), fr = __shfl_down(pc[r].l, 1, 64);
                const double pxm = laneL ? fl : (double)phi((wrap && iL == flo0) ? fhi0 : iL - 1, j, k, 0);
                const double pxp = laneR ? fr : (double)phi((wrap && iR == fhi0) ? flo0 : iR + 1, j, k, 0);
                const D2 pym = r == 0 ? pS : pc[0], pyp = r == 0 ? pc[1] : pN;
                double bxl, bxc, bxr, byml, bymr, bypl, bypr, bzml, bzmr, bzpl, bzpr;        // x faces: left of iL, between, right of iR
                if (BMODE == 1) {
                    const double gl = __shfl_up(sc[r].r, 1, 64), gr = __shfl_down(sc[r].l, 1, 64);
                    const double sxm = laneL ? gl : (double)S(iL - 1, j, k, sig_comp), sxp = laneR ? gr : (double)S(iR + 1, j, k, sig_comp);
                    const D2 sym = r == 0 ? sS : sc[0], syp = r == 0 ? sc[1] : sN;
                    bxl = bco(sxm, sc[r].l); bxc = bco(sc[r].l, sc[r].r); bxr = bco(sc[r].r, sxp);
                    byml = bco(sym.l, sc[r].l); bymr = bco(sym.r, sc[r].r); bypl = bco(sc[r].l, syp.l); bypr = bco(sc[r].r, syp.r);
                    bzml = bco(sb[r].l, sc[r].l); bzmr = bco(sb[r].r, sc[r].r); bzpl = bco(sc[r].l, sa[r].l); bzpr = bco(sc[r].r, sa[r].r);
                } else {
                    bxl = bxc = bxr = bu.v[0]; byml = bymr = bypl = bypr = bu.v[1]; bzml = bzmr = bzpl = bzpr = bu.v[2];
                }
                {   // cell iL
                    const double p0 = pc[r].l;
                    const double y = 0.0 - dhx * (bxc * (pc[r].r - p0) - bxl * (p0 - pxm)) - dhy * (bypl * (pyp.l - p0) - byml * (p0 - pym.l))
                                   - dhz * (bzpl * (pa[r].l - p0) - bzml * (p0 - pb[r].l));
                    const double o = rr.l - y;
                    if (RESTRICT) s += o; else { crse(iL, j, k, 0) = o; mx = fmax(mx, norm_term(o)); }
                }
                {   // cell iR
                    const double p0 = pc[r].r;
                    const double y = 0.0 - dhx * (bxr * (pxp - p0) - bxc * (p0 - pc[r].l)) - dhy * (bypr * (pyp.r - p0) - bymr * (p0 - pym.r))
                                   - dhz * (bzpr * (pa[r].r - p0) - bzmr * (p0 - pb[r].r));
                    const double o = rr.r - y;
                    if (RESTRICT) s += o; else { crse(iR, j, k, 0) = o; mx = fmax(mx, norm_term(o)); }
                }
            }
            for (int r = 0; r < 2; ++r) { pb[r] = pc[r]; pc[r] = pa[r]; if (BMODE == 1) { sb[r] = sc[r]; sc[r] = sa[r]; } }
        }
        if (RESTRICT) crse(I, J, K, 0) = 0.125 * s;
    }
    if (!RESTRICT && normout) norm_commit(mx, normout);
}

bool abec_resid_restrict_ok(const AbecCoef& c, const MultiFab& phi, const MultiFab& rhs);
// the pair-marching residual kernels read phi's periodic images from its valid cells on this level (IAMRX_RESID_WRAP, 1)
static bool abec_resid_wrap(const Geometry& g, const Layout& l) { return tune("RESID_WRAP", 1) != 0 && periodic_wrap_ok(g, l, 4); }

// abec_residual(g, c, out, phi, rhs) / abec_resid_restrict will read no ghost cell of phi: the caller may skip the ghost fill in front of it
bool abec_residual_reads_no_ghosts(const Geometry& g, const AbecCoef& c, const MultiFab& out, const MultiFab& phi, const MultiFab& rhs, bool restrict_form)
{
    if (!abec_resid_wrap(g, *phi.layout) || !abec_resid_restrict_ok(c, phi, rhs) || c.tensor) return false;
    if (restrict_form) return true;
    return tune("RESID_PAIRS", 1) != 0 && phi.layout->coarsenable(2, 1);
}

bool abec_resid_restrict_ok(const AbecCoef& c, const MultiFab& phi, const MultiFab& rhs)
{
    if (tune("RESID_RESTRICT", 1) == 0 || tune("ABEC_SIG", 1) == 0) return false;
    if (phi.ncomp != 1 || c.tensor || c.tensor_eta || (c.a && c.alpha != 0.0) || phi.ngrow < 1) return false;
    if (c.sig) return c.sig->ngrow >= 1;
    return c.b_uniform && c.b[0]->ncomp == 1;
}

// crse = restriction of (rhs - A phi); see k_abec_resid_restrict.  crse: the coarsened layout of phi's
void abec_resid_restrict(const Geometry& g, const AbecCoef& c, MultiFab& crse, const MultiFab& phi, const MultiFab& rhs)
{
    IAMRX_ASSERT(abec_resid_restrict_ok(c, phi, rhs));
    if (crse.nlocal() == 0) return;
    auto& ctx = Context::get();
    const Layout& l = *crse.layout;
    const int wrap = abec_resid_wrap(g, *phi.layout) ? 1 : 0;
    Tiling t = level_tiling(l, cell_type(), 0, (int)tune("RESID_RESTRICT_TZ", 8));
    const double dhx = c.beta / (g.dx[0] * g.dx[0]), dhy = c.beta / (g.dx[1] * g.dx[1]), dhz = c.beta / (g.dx[2] * g.dx[2]);
    BUni bu;
    for (int d = 0; d < 3; ++d) bu.v[d] = c.bu[d];
    if (c.sig)
        hipLaunchKernelGGL((k_abec_resid_restrict<1, true>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, crse.d_tab, phi.d_tab, rhs.d_tab, c.sig->d_tab,
                           dhx, dhy, dhz, c.sig_comp, c.sig_scale, bu, nullptr, wrap);
    else
        hipLaunchKernelGGL((k_abec_resid_restrict<2, true>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, crse.d_tab, phi.d_tab, rhs.d_tab, nullptr,
                           dhx, dhy, dhz, 0, 1.0, bu, nullptr, wrap);
}

// the fine residual out = rhs - A phi by the same march (k_abec_resid_restrict<., false>): levels whose boxes coarsen by 2
static bool abec_residual_pairs(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& phi, const MultiFab& rhs, unsigned long long* d_norm)
{
    if (tune("RESID_PAIRS", 1) == 0 || !abec_resid_restrict_ok(c, phi, rhs)) return false;
    const Layout& fl = *phi.layout;
    if (!fl.coarsenable(2, 1)) return false;
    LayoutP cl = fl.coarsened(2);
    auto& ctx = Context::get();
    const int wrap = abec_resid_wrap(g, fl) ? 1 : 0;
    Tiling t = level_tiling(*cl, cell_type(), 0, (int)tune("RESID_RESTRICT_TZ", 8));
    const double dhx = c.beta / (g.dx[0] * g.dx[0]), dhy = c.beta / (g.dx[1] * g.dx[1]), dhz = c.beta / (g.dx[2] * g.dx[2]);
    BUni bu;
    for (int d = 0; d < 3; ++d) bu.v[d] = c.bu[d];
    if (c.sig)
        hipLaunchKernelGGL((k_abec_resid_restrict<1, false>), t.grid(), Tiling::block(), 0, ctx.stream, t, cl->d_boxes, out.d_tab, phi.d_tab, rhs.d_tab, c.sig->d_tab,
                           dhx, dhy, dhz, c.sig_comp, c.sig_scale, bu, d_norm, wrap);
    else
        hipLaunchKernelGGL((k_abec_resid_restrict<2, false>), t.grid(), Tiling::block(), 0, ctx.stream, t, cl->d_boxes, out.d_tab, phi.d_tab, rhs.d_tab, nullptr,
                           dhx, dhy, dhz, 0, 1.0, bu, d_norm, wrap);
    return true;
}

// ---------------------------------------------------------------------------- bottom solve of a small level on the device
// CellMG::bottom_solve in ONE launch of ONE workgroup for a level that is a single box of at most BOT_NT cells spanning the domain:
// BiCGStab as CellMG::bicgstab drives it from the host (mlmg.hip: same recurrences, same breakdown / convergence tests), but with the
// vectors in registers (one cell per thread), the ghost-cell work array in LDS and every dot product / max norm reduced inside the
// workgroup -- no host synchronisation (5 per Krylov iteration in the host-driven form), no launches (~14 per iteration); then the
// nub (converged) or nuf (fallback) red-black sweeps.  On MI355X the host-driven bottom of a 256^3 hierarchy cost 0.47 ms of every
// 2.6 ms V-cycle.
constexpr int BOT_NT = 512;
struct BotBC { int per[3]; int bct[6]; double c[6][5]; };

// (NW = the wavefronts that hold cells: the partial sums are added in wavefront order -- k_abec_tail runs the 512-cell solve in a
// 1024-thread workgroup and adds the same BOT_NT / 64 terms)
template <int NW = BOT_NT / 64>
__device__ __forceinline__ double bot_sum(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}
template <int NW = BOT_NT / 64>
__device__ __forceinline__ double bot_max(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < NW; ++w) s = fmax(s, red[w]);
    return s;
}

// ghost cells of the faces of W ((nx+2)(ny+2)(nz+2), interior filled): periodic images, then the homogeneous domain BCs (k_abec_bc)
template <int NT = BOT_NT>
__device__ __forceinline__ void bot_fill_ghosts(double* W, int nx, int ny, int nz, const BotBC& bc)
{
    const int sx = 1, sy = nx + 2, sz = (nx + 2) * (ny + 2);
    const int nfx = ny * nz, nfy = nx * nz, nfz = nx * ny;
    __syncthreads();
    for (int q = threadIdx.x; q < 2 * (nfx + nfy + nfz); q += NT) {
        int d, side, a, b2, r = q;
        if (r < 2 * nfx) { d = 0; side = r / nfx; r %= nfx; a = r % ny; b2 = r / ny; }
        else if (r < 2 * (nfx + nfy)) { r -= 2 * nfx; d = 1; side = r / nfy; r %= nfy; a = r % nx; b2 = r / nx; }
        else { r -= 2 * (nfx + nfy); d = 2; side = r / nfz; r %= nfz; a = r % nx; b2 = r / nx; }
        const int n = d == 0 ? nx : (d == 1 ? ny : nz), st = d == 0 ? sx : (d == 1 ? sy : sz);
        // offset of the ghost cell: transverse coordinates (1-based interior) a, b2
        int base;
        if (d == 0) base = (a + 1) * sy + (b2 + 1) * sz;
        else if (d == 1) base = (a + 1) * sx + (b2 + 1) * sz;
        else base = (a + 1) * sx + (b2 + 1) * sy;
        const int gi = side == 0 ? 0 : n + 1;            // ghost index along d
        const int s = side == 0 ? 1 : -1;
        double v = 0.0;
        if (bc.per[d]) v = W[base + (side == 0 ? n : 1) * st];
        else {
            const int t = bc.bct[2 * d + side];
            const double* c = bc.c[2 * d + side];
            if (t == lo_neumann) v = W[base + (gi + s) * st];
            else if (t == lo_reflect_odd) v = -W[base + (gi + s) * st];
            else {
                const int NX = (int)c[4];
                for (int q2 = 1; q2 < NX; ++q2) v += W[base + (gi + q2 * s) * st] * c[q2];
            }
        }
        W[base + gi * st] = v;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(BOT_NT) k_abec_bottom(BoxD b, const FabD* __restrict__ cort, const FabD* __restrict__ rest,
    const FabD* __restrict__ at, const FabD* __restrict__ bxt, const FabD* __restrict__ byt, const FabD* __restrict__ bzt,
    double alpha, double dhx, double dhy, double dhz, BotBC bc, GsrbBC gb, int singular, double eps_rel, int maxiter, int nub, int nuf,
    double omega_gs, int* __restrict__ iters_out)
{
    __shared__ double W[(8 + 2) * (8 + 2) * (8 + 2) * 2];      // up to 512 cells in any box shape with sides <= 8 ... 64: see host check
    __shared__ double red[BOT_NT / 64];
    const int nx = b.len(0), ny = b.len(1), nz = b.len(2), nc = nx * ny * nz;
    const int sy = nx + 2, sz = (nx + 2) * (ny + 2), nw = sz * (nz + 2);
    const int tid = threadIdx.x;
    const bool on = tid < nc;
    const int li = on ? tid % nx : 0, lj = on ? (tid / nx) % ny : 0, lk = on ? tid / (nx * ny) : 0;
    const int i = b.lo[0] + li, j = b.lo[1] + lj, k = b.lo[2] + lk;
    const int w0 = (li + 1) + (lj + 1) * sy + (lk + 1) * sz;
    const FabD cor = cort[0], res = rest[0], bX = bxt[0], bY = byt[0], bZ = bzt[0];
    const bool has_a = at != nullptr && alpha != 0.0;
    // coefficients of the own cell
    double bxm = 0, bxp = 0, bym = 0, byp = 0, bzm = 0, bzp = 0, aa = 0, rhs0 = 0;
    if (on) {
        bxm = bX(i, j, k, 0); bxp = bX(i + 1, j, k, 0); bym = bY(i, j, k, 0); byp = bY(i, j + 1, k, 0); bzm = bZ(i, j, k, 0); bzp = bZ(i, j, k + 1, 0);
        if (has_a) aa = alpha * at[0](i, j, k, 0);
        rhs0 = res(i, j, k, 0);
    }
    for (int q = tid; q < nw; q += BOT_NT) W[q] = 0.0;
    // y = A x for the vector held one cell per thread (k_abec_residual's expression)
    auto apply = [&](double xv) -> double {
        __syncthreads();
        if (on) W[w0] = xv;
        bot_fill_ghosts(W, nx, ny, nz, bc);
        if (!on) return 0.0;
        const double p0 = xv;
        return (has_a ? aa * p0 : 0.0)
            - dhx * (bxp * (W[w0 + 1] - p0) - bxm * (p0 - W[w0 - 1]))
            - dhy * (byp * (W[w0 + sy] - p0) - bym * (p0 - W[w0 - sy]))
            - dhz * (bzp * (W[w0 + sz] - p0) - bzm * (p0 - W[w0 - sz]));
    };
    double bb = rhs0;
    if (singular) bb -= bot_sum(on ? rhs0 : 0.0, red) / (double)nc;
    if (!on) bb = 0.0;
    double x = 0.0, r = bb, p = 0.0, v = 0.0;
    const double rh = r;
    const double rnorm0 = bot_max(fabs(r), red);
    double rnorm = rnorm0;
    int ret = 0, nit = 0;
    if (rnorm0 != 0.0) {
        double rho_1 = 0.0, alph = 0.0, omg = 0.0;
        for (nit = 1; nit <= maxiter; ++nit) {
            const double rho = bot_sum(rh * r, red);
            if (rho == 0.0) { ret = 1; break; }
            if (nit == 1) p = r;
            else {
                const double beta = (rho / rho_1) * (alph / omg);
                p = p - omg * v;
                p = r + beta * p;
            }
            v = apply(p);
            const double rhTv = bot_sum(rh * v, red);
            if (rhTv != 0.0) alph = rho / rhTv; else { ret = 2; break; }
            x = x + alph * p;
            const double s = r - alph * v;
            rnorm = bot_max(fabs(s), red);
            if (rnorm < eps_rel * rnorm0) break;
            const double t = apply(s);
            const double tt = bot_sum(t * t, red), ts = bot_sum(t * s, red);
            if (tt != 0.0) omg = ts / tt; else { ret = 3; break; }
            x = x + omg * s;
            r = s - omg * t;
            rnorm = bot_max(fabs(r), red);
            if (rnorm < eps_rel * rnorm0) break;
            if (omg == 0.0) { ret = 4; break; }
            rho_1 = rho;
        }
        if (ret == 0 && rnorm > eps_rel * rnorm0) ret = 8;
        if (!((ret == 0 || ret == 8) && rnorm < rnorm0)) x = 0.0;
    }
    if (tid == 0 && iters_out) atomicAdd(iters_out, nit);
    // smoothing: a failed Krylov solve is replaced by nuf sweeps from zero, then nub / nuf sweeps (CellMG::bottom_solve)
    int nsw = ret == 0 ? nub : nuf;
    if (ret != 0) { x = 0.0; nsw += nuf; }
    const double gamma = aa + dhx * (bxm + bxp) + dhy * (bym + byp) + dhz * (bzm + bzp);
    const double cf0 = (i == gb.dlo[0]) ? gb.cflo[0][0] : 0.0, cf3 = (i == gb.dhi[0]) ? gb.cfhi[0][0] : 0.0;
    const double cf1 = (j == gb.dlo[1]) ? gb.cflo[0][1] : 0.0, cf4 = (j == gb.dhi[1]) ? gb.cfhi[0][1] : 0.0;
    const double cf2 = (k == gb.dlo[2]) ? gb.cflo[0][2] : 0.0, cf5 = (k == gb.dhi[2]) ? gb.cfhi[0][2] : 0.0;
    const double g_m_d = gamma - (dhx * (bxm * cf0 + bxp * cf3) + dhy * (bym * cf1 + byp * cf4) + dhz * (bzm * cf2 + bzp * cf5));
    for (int sw = 0; sw < nsw; ++sw)
        for (int rb = 0; rb < 2; ++rb) {
            __syncthreads();
            if (on) W[w0] = x;
            bot_fill_ghosts(W, nx, ny, nz, bc);
            if (on && ((li & 1) == ((b.lo[0] + j + k + rb) & 1))) {
                const double rho = dhx * (bxm * W[w0 - 1] + bxp * W[w0 + 1]) + dhy * (bym * W[w0 - sy] + byp * W[w0 + sy])
                                 + dhz * (bzm * W[w0 - sz] + bzp * W[w0 + sz]);
                const double resid = rhs0 - (gamma * x - rho);
                x = x + omega_gs / g_m_d * resid;
            }
        }
    if (on) cor(i, j, k, 0) = x;
}

// cftab != nullptr: the level belongs to a refined AMR level (MLLinOp::setCoarseFineBC): faces of the box that are not domain faces (or
// are faces of a periodic direction the box does not span) carry homogeneous coarse/fine Dirichlet data with the level's CfTab weights
bool abec_bottom_device_ok(const Geometry& g, const Layout& l, const DomainBC* bcs, int nbc, int ncomp, bool cf)
{
    const bool enabled = tune("MG_DEVICE_BOTTOM", 1) != 0;
    // the answer must be the same on every rank (it decides the depth of the hierarchy): global information only.  A rank that does not
    // own the box has nothing to do in the bottom solve (no collectives in it)
    if (!enabled || ncomp != 1 || nbc != 1 || l.boxes.size() != 1) return false;
    const BoxD b = l.boxes[0];
    for (int d = 0; d < 3; ++d) {
        if (b.len(d) > 8) return false;
        const bool spans = b.lo[d] == g.domain.lo[d] && b.hi[d] == g.domain.hi[d];
        if (!cf && !spans) return false;
        if (g.periodic[d]) continue;
        for (int side = 0; side < 2; ++side) {
            const bool on_dom = side == 0 ? b.lo[d] == g.domain.lo[d] : b.hi[d] == g.domain.hi[d];
            if (!on_dom) continue;
            const int t = side == 0 ? bcs[0].lo[d] : bcs[0].hi[d];
            if (t != lo_neumann && t != lo_dirichlet && t != lo_reflect_odd) return false;
        }
    }
    return b.npts() <= BOT_NT;
}

// ghost formulas of a box's six faces for the single-workgroup kernels (BotBC: the fills inside LDS; GsrbBC: the first-interior-cell weights,
// relative to the faces of the box)
static void make_bot_bc(const Geometry& g, const BoxD& b, const DomainBC& bc, const CfTab* cftab, BotBC& bb, GsrbBC& gb)
{
    gb.nbc = 1;
    for (int d = 0; d < 3; ++d) {
        const bool spans = b.lo[d] == g.domain.lo[d] && b.hi[d] == g.domain.hi[d];
        bb.per[d] = g.periodic[d] && spans;
        gb.dlo[d] = b.lo[d]; gb.dhi[d] = b.hi[d];
        for (int side = 0; side < 2; ++side) {
            const bool on_dom = !g.periodic[d] && (side == 0 ? b.lo[d] == g.domain.lo[d] : b.hi[d] == g.domain.hi[d]);
            double cc[4] = {0.0, 0.0, 0.0, 0.0};
            int NX = 0, type = lo_periodic;
            double first = 0.0;
            if (bb.per[d]) { type = lo_periodic; }
            else if (on_dom) {
                type = side == 0 ? bc.lo[d] : bc.hi[d];
                dirichlet_coefs(g.domain.len(d), bc.maxorder, cc, NX);
                first = type == lo_neumann ? 1.0 : (type == lo_reflect_odd ? -1.0 : (NX >= 2 ? cc[1] : 0.0));
            } else {                 // coarse/fine face: Dirichlet point half a coarse cell behind the face (CfTab), homogeneous in the correction
                IAMRX_ASSERT(cftab != nullptr);
                type = lo_dirichlet;
                NX = std::min(b.len(d) + 1, cftab->maxorder);
                for (int q = 0; q < NX; ++q) cc[q] = cftab->c[d][NX - 2][q];
                first = cc[1];
            }
            bb.bct[2 * d + side] = type;
            for (int q = 0; q < 4; ++q) bb.c[2 * d + side][q] = cc[q];
            bb.c[2 * d + side][4] = NX;
            for (int n = 0; n < 3; ++n) (side == 0 ? gb.cflo[n][d] : gb.cfhi[n][d]) = first;
        }
    }
}

void abec_bottom_solve(const Geometry& g, const AbecCoef& c, MultiFab& cor, const MultiFab& res, const DomainBC& bc, bool singular,
                       double eps_rel, int maxiter, int nub, int nuf, double omega, int* d_iters, const CfTab* cftab)
{
    const Layout& l = *cor.layout;
    IAMRX_ASSERT(abec_bottom_device_ok(g, l, &bc, 1, cor.ncomp, cftab != nullptr) && cor.ngrow >= 1);
    if (l.nlocal() == 0) return;
    const BoxD b = l.boxes[0];
    BotBC bb;
    GsrbBC gb;
    make_bot_bc(g, b, bc, cftab, bb, gb);
    const double dhx = c.beta / (g.dx[0] * g.dx[0]), dhy = c.beta / (g.dx[1] * g.dx[1]), dhz = c.beta / (g.dx[2] * g.dx[2]);
    hipLaunchKernelGGL(k_abec_bottom, dim3(1), dim3(BOT_NT), 0, Context::get().stream, b, cor.d_tab, res.d_tab,
                       c.a ? c.a->d_tab : nullptr, c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab, c.alpha, dhx, dhy, dhz, bb,
                       gb, singular ? 1 : 0, eps_rel, maxiter, nub, nuf, omega, d_iters);
}

// ---------------------------------------------------------------------------- the coarse tail of a V-cycle in ONE launch
// The last two levels of a cell-centred hierarchy -- a level F of at most 16^3 cells and the bottom level C (its coarsening by 2, at most
// 8^3 = 512 cells: k_abec_bottom's) -- in one launch of one 1024-thread workgroup: nu1 red-black sweeps on F from zero, residual,
// restriction, the bottom solve (BiCGStab + sweeps, k_abec_bottom's code on the first 512 threads), prolongation, nu2 sweeps.  It replaces
// 14 launches of ~5 us (4 colour passes, ghost fill, residual, restriction, fill, bottom, prolongation, 4 colour passes) per V-cycle; a
// thread owns up to four cells of F (coefficients, right-hand side and correction in registers), the correction with its ghost cells
// lives in LDS (18^3 doubles), ghost cells are filled there (bot_fill_ghosts: periodic images, Neumann, reflect-odd, Dirichlet of any
// order -- k_abec_bc's formulas).  Every expression is that of the kernel it replaces (k_abec_gsrb1, k_abec_residual, k_cc_restrict,
// k_abec_bottom, cc_prolong_add): the same doubles (tests/test_gpu_sensitivity.py: MG_TAIL_FUSED = 0 / 1 bit for bit).
constexpr int TAIL_NT = 1024, TAIL_CPT = 4, TAIL_NF = 16;
struct TailLevel { BoxD b; double dhx, dhy, dhz; BotBC bc; GsrbBC gb; };
struct TailTabs { const FabD *cor, *res, *a, *bx, *by, *bz; };

__global__ void __launch_bounds__(TAIL_NT) k_abec_tail(TailLevel F, TailLevel C, TailTabs tf, TailTabs tc, double alpha, int uni, BUni bu,
    int nu1, int nu2, double omega_gs, int singular, double eps_rel, int maxiter, int nub, int nuf, int* __restrict__ iters_out)
{
    constexpr int NFA = (TAIL_NF + 1) * TAIL_NF * TAIL_NF;          // a face-coefficient array of F
    __shared__ double WF[(TAIL_NF + 2) * (TAIL_NF + 2) * (TAIL_NF + 2)];
    __shared__ double WC[(8 + 2) * (8 + 2) * (8 + 2)];
    __shared__ double BF[3 * NFA];                                   // F's face coefficients (arrays form): 24 doubles per thread otherwise
    __shared__ double red[TAIL_NT / 64];
    const int tid = threadIdx.x;
    // ---------------------------------------------------------------- level F: the cells of this thread
    const BoxD bF = F.b;
    const int nxF = bF.len(0), nyF = bF.len(1), nzF = bF.len(2), ncF = nxF * nyF * nzF;
    const int syF = nxF + 2, szF = (nxF + 2) * (nyF + 2), nwF = szF * (nzF + 2);
    const FabD corF = tf.cor[0], resF = tf.res[0];
    const bool has_aF = tf.a != nullptr && alpha != 0.0;
    // face coefficients of F in LDS: bX(li, lj, lk) at li + (nxF + 1) (lj + nyF lk), bY at NFA + li + nxF (lj + (nyF + 1) lk), bZ at 2 NFA + li + nxF (lj + nyF lk)
    const int sxy = (nxF + 1), sxz = (nxF + 1) * nyF, syy = nxF, syz = nxF * (nyF + 1), szy = nxF, szz = nxF * nyF;
    if (!uni) {
        const FabD bX = tf.bx[0], bY = tf.by[0], bZ = tf.bz[0];
        for (int q = tid; q < (nxF + 1) * nyF * nzF; q += TAIL_NT) BF[q] = bX(bF.lo[0] + q % (nxF + 1), bF.lo[1] + (q / (nxF + 1)) % nyF, bF.lo[2] + q / ((nxF + 1) * nyF), 0);
        for (int q = tid; q < nxF * (nyF + 1) * nzF; q += TAIL_NT) BF[NFA + q] = bY(bF.lo[0] + q % nxF, bF.lo[1] + (q / nxF) % (nyF + 1), bF.lo[2] + q / (nxF * (nyF + 1)), 0);
        for (int q = tid; q < nxF * nyF * (nzF + 1); q += TAIL_NT) BF[2 * NFA + q] = bZ(bF.lo[0] + q % nxF, bF.lo[1] + (q / nxF) % nyF, bF.lo[2] + q / (nxF * nyF), 0);
    }
    __syncthreads();
    bool onF[TAIL_CPT];
    int wF[TAIL_CPT], pF[TAIL_CPT], cF[TAIL_CPT], oX[TAIL_CPT], oY[TAIL_CPT], oZ[TAIL_CPT];
    double x[TAIL_CPT], rF[TAIL_CPT], aaF[TAIL_CPT], gamF[TAIL_CPT], gmdF[TAIL_CPT];
    // the six face coefficients of cell m: constants or LDS
    auto bco = [&](int m, int q) -> double {
        if (uni) return bu.v[q >> 1];
        switch (q) {
        case 0: return BF[oX[m]];
        case 1: return BF[oX[m] + 1];
        case 2: return BF[NFA + oY[m]];
        case 3: return BF[NFA + oY[m] + syy];
        case 4: return BF[2 * NFA + oZ[m]];
        default: return BF[2 * NFA + oZ[m] + szz];
        }
    };
#pragma unroll
    for (int m = 0; m < TAIL_CPT; ++m) {
        const int idx = tid + m * TAIL_NT;
        onF[m] = idx < ncF;
        const int li = onF[m] ? idx % nxF : 0, lj = onF[m] ? (idx / nxF) % nyF : 0, lk = onF[m] ? idx / (nxF * nyF) : 0;
        const int i = bF.lo[0] + li, j = bF.lo[1] + lj, k = bF.lo[2] + lk;
        wF[m] = (li + 1) + (lj + 1) * syF + (lk + 1) * szF;
        oX[m] = li + sxy * lj + sxz * lk; oY[m] = li + syy * lj + syz * lk; oZ[m] = li + szy * lj + szz * lk;
        pF[m] = (i + j + k) & 1;                                               // the cell is updated in the colour pass rb = pF
        cF[m] = (li >> 1) + 1 + ((lj >> 1) + 1) * (C.b.len(0) + 2) + ((lk >> 1) + 1) * (C.b.len(0) + 2) * (C.b.len(1) + 2);      // its coarse cell in WC
        x[m] = 0.0; rF[m] = 0.0; aaF[m] = 0.0;
        if (onF[m]) {
            if (has_aF) aaF[m] = alpha * tf.a[0](i, j, k, 0);
            rF[m] = resF(i, j, k, 0);
        }
        const double bxm = bco(m, 0), bxp = bco(m, 1), bym = bco(m, 2), byp = bco(m, 3), bzm = bco(m, 4), bzp = bco(m, 5);
        gamF[m] = aaF[m] + F.dhx * (bxm + bxp) + F.dhy * (bym + byp) + F.dhz * (bzm + bzp);
        const double cf0 = (i == F.gb.dlo[0]) ? F.gb.cflo[0][0] : 0.0, cf3 = (i == F.gb.dhi[0]) ? F.gb.cfhi[0][0] : 0.0;
        const double cf1 = (j == F.gb.dlo[1]) ? F.gb.cflo[0][1] : 0.0, cf4 = (j == F.gb.dhi[1]) ? F.gb.cfhi[0][1] : 0.0;
        const double cf2 = (k == F.gb.dlo[2]) ? F.gb.cflo[0][2] : 0.0, cf5 = (k == F.gb.dhi[2]) ? F.gb.cfhi[0][2] : 0.0;
        gmdF[m] = gamF[m] - (F.dhx * (bxm * cf0 + bxp * cf3) + F.dhy * (bym * cf1 + byp * cf4) + F.dhz * (bzm * cf2 + bzp * cf5));
    }
    for (int q = tid; q < nwF; q += TAIL_NT) WF[q] = 0.0;
    // the correction with its ghost cells in LDS (every wavefront is past its reads of the previous contents: barrier first)
    auto publishF = [&]() {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < TAIL_CPT; ++m) if (onF[m]) WF[wF[m]] = x[m];
        bot_fill_ghosts<TAIL_NT>(WF, nxF, nyF, nzF, F.bc);
    };
    // nsw red-black sweeps (k_abec_gsrb1's update)
    auto sweepsF = [&](int nsw) {
        for (int sw = 0; sw < nsw; ++sw)
            for (int rb = 0; rb < 2; ++rb) {
                publishF();
#pragma unroll
                for (int m = 0; m < TAIL_CPT; ++m)
                    if (onF[m] && ((pF[m] + rb) & 1) == 0) {
                        const int w = wF[m];
                        const double bxm = bco(m, 0), bxp = bco(m, 1), bym = bco(m, 2), byp = bco(m, 3), bzm = bco(m, 4), bzp = bco(m, 5);
                        const double rho = F.dhx * (bxm * WF[w - 1] + bxp * WF[w + 1]) + F.dhy * (bym * WF[w - syF] + byp * WF[w + syF])
                                         + F.dhz * (bzm * WF[w - szF] + bzp * WF[w + szF]);
                        const double resid = rF[m] - (gamF[m] * x[m] - rho);
                        x[m] = x[m] + omega_gs / gmdF[m] * resid;
                    }
            }
    };
    sweepsF(nu1);
    // ---------------------------------------------------------------- residual of F (k_abec_residual), restricted onto C (k_cc_restrict)
    publishF();
    double rr[TAIL_CPT];
#pragma unroll
    for (int m = 0; m < TAIL_CPT; ++m) {
        rr[m] = 0.0;
        if (onF[m]) {
            const int w = wF[m];
            const double p0 = x[m];
            const double bxm = bco(m, 0), bxp = bco(m, 1), bym = bco(m, 2), byp = bco(m, 3), bzm = bco(m, 4), bzp = bco(m, 5);
            const double ax = has_aF ? aaF[m] * p0 : 0.0;
            const double y = ax
                - F.dhx * (bxp * (WF[w + 1] - p0) - bxm * (p0 - WF[w - 1]))
                - F.dhy * (byp * (WF[w + syF] - p0) - bym * (p0 - WF[w - syF]))
                - F.dhz * (bzp * (WF[w + szF] - p0) - bzm * (p0 - WF[w - szF]));
            rr[m] = rF[m] - y;
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TAIL_CPT; ++m) if (onF[m]) WF[wF[m]] = rr[m];
    __syncthreads();
    // ---------------------------------------------------------------- level C: k_abec_bottom's solve on the first nc threads
    const BoxD b = C.b;
    const int nx = b.len(0), ny = b.len(1), nz = b.len(2), nc = nx * ny * nz;
    const int sy = nx + 2, sz = (nx + 2) * (ny + 2), nw = sz * (nz + 2);
    const bool on = tid < nc;
    const int li = on ? tid % nx : 0, lj = on ? (tid / nx) % ny : 0, lk = on ? tid / (nx * ny) : 0;
    const int i = b.lo[0] + li, j = b.lo[1] + lj, k = b.lo[2] + lk;
    const int w0 = (li + 1) + (lj + 1) * sy + (lk + 1) * sz;
    const bool has_a = tc.a != nullptr && alpha != 0.0;
    double bxm = 0, bxp = 0, bym = 0, byp = 0, bzm = 0, bzp = 0, aa = 0, rhs0 = 0;
    if (on) {
        if (uni) { bxm = bxp = bu.v[0]; bym = byp = bu.v[1]; bzm = bzp = bu.v[2]; }
        else {
            const FabD bX = tc.bx[0], bY = tc.by[0], bZ = tc.bz[0];
            bxm = bX(i, j, k, 0); bxp = bX(i + 1, j, k, 0); bym = bY(i, j, k, 0); byp = bY(i, j + 1, k, 0); bzm = bZ(i, j, k, 0); bzp = bZ(i, j, k + 1, 0);
        }
        if (has_a) aa = alpha * tc.a[0](i, j, k, 0);
        // the restricted residual: k_cc_restrict's sum over the 2 x 2 x 2 fine cells, in its order
        double s = 0.0;
        for (int kr = 0; kr < 2; ++kr)
            for (int jr = 0; jr < 2; ++jr) {
                const int wf = (2 * li + 1) + (2 * lj + jr + 1) * syF + (2 * lk + kr + 1) * szF;
                s += WF[wf];
                s += WF[wf + 1];
            }
        rhs0 = 0.125 * s;
    }
    const double dhx = C.dhx, dhy = C.dhy, dhz = C.dhz;
    const BotBC& bc = C.bc;
    const GsrbBC& gb = C.gb;
    for (int q = tid; q < nw; q += TAIL_NT) WC[q] = 0.0;
    double* const W = WC;
    auto apply = [&](double xv) -> double {
        __syncthreads();
        if (on) W[w0] = xv;
        bot_fill_ghosts<TAIL_NT>(W, nx, ny, nz, bc);
        if (!on) return 0.0;
        const double p0 = xv;
        return (has_a ? aa * p0 : 0.0)
            - dhx * (bxp * (W[w0 + 1] - p0) - bxm * (p0 - W[w0 - 1]))
            - dhy * (byp * (W[w0 + sy] - p0) - bym * (p0 - W[w0 - sy]))
            - dhz * (bzp * (W[w0 + sz] - p0) - bzm * (p0 - W[w0 - sz]));
    };
    double bb = rhs0;
    if (singular) bb -= bot_sum(on ? rhs0 : 0.0, red) / (double)nc;
    if (!on) bb = 0.0;
    double xc = 0.0, r = bb, p = 0.0, v = 0.0;
    const double rh = r;
    const double rnorm0 = bot_max(fabs(r), red);
    double rnorm = rnorm0;
    int ret = 0, nit = 0;
    if (rnorm0 != 0.0) {
        double rho_1 = 0.0, alph = 0.0, omg = 0.0;
        for (nit = 1; nit <= maxiter; ++nit) {
            const double rho = bot_sum(rh * r, red);
            if (rho == 0.0) { ret = 1; break; }
            if (nit == 1) p = r;
            else {
                const double beta = (rho / rho_1) * (alph / omg);
                p = p - omg * v;
                p = r + beta * p;
            }
            v = apply(p);
            const double rhTv = bot_sum(rh * v, red);
            if (rhTv != 0.0) alph = rho / rhTv; else { ret = 2; break; }
            xc = xc + alph * p;
            const double s = r - alph * v;
            rnorm = bot_max(fabs(s), red);
            if (rnorm < eps_rel * rnorm0) break;
            const double t = apply(s);
            const double tt = bot_sum(t * t, red), ts = bot_sum(t * s, red);
            if (tt != 0.0) omg = ts / tt; else { ret = 3; break; }
            xc = xc + omg * s;
            r = s - omg * t;
            rnorm = bot_max(fabs(r), red);
            if (rnorm < eps_rel * rnorm0) break;
            if (omg == 0.0) { ret = 4; break; }
            rho_1 = rho;
        }
        if (ret == 0 && rnorm > eps_rel * rnorm0) ret = 8;
        if (!((ret == 0 || ret == 8) && rnorm < rnorm0)) xc = 0.0;
    }
    if (tid == 0 && iters_out) atomicAdd(iters_out, nit);
    int nsw = ret == 0 ? nub : nuf;
    if (ret != 0) { xc = 0.0; nsw += nuf; }
    {
        const double gamma = aa + dhx * (bxm + bxp) + dhy * (bym + byp) + dhz * (bzm + bzp);
        const double cf0 = (i == gb.dlo[0]) ? gb.cflo[0][0] : 0.0, cf3 = (i == gb.dhi[0]) ? gb.cfhi[0][0] : 0.0;
        const double cf1 = (j == gb.dlo[1]) ? gb.cflo[0][1] : 0.0, cf4 = (j == gb.dhi[1]) ? gb.cfhi[0][1] : 0.0;
        const double cf2 = (k == gb.dlo[2]) ? gb.cflo[0][2] : 0.0, cf5 = (k == gb.dhi[2]) ? gb.cfhi[0][2] : 0.0;
        const double g_m_d = gamma - (dhx * (bxm * cf0 + bxp * cf3) + dhy * (bym * cf1 + byp * cf4) + dhz * (bzm * cf2 + bzp * cf5));
        for (int sw = 0; sw < nsw; ++sw)
            for (int rb = 0; rb < 2; ++rb) {
                __syncthreads();
                if (on) W[w0] = xc;
                bot_fill_ghosts<TAIL_NT>(W, nx, ny, nz, bc);
                if (on && ((li & 1) == ((b.lo[0] + j + k + rb) & 1))) {
                    const double rho = dhx * (bxm * W[w0 - 1] + bxp * W[w0 + 1]) + dhy * (bym * W[w0 - sy] + byp * W[w0 + sy])
                                     + dhz * (bzm * W[w0 - sz] + bzp * W[w0 + sz]);
                    const double resid = rhs0 - (gamma * xc - rho);
                    xc = xc + omega_gs / g_m_d * resid;
                }
            }
    }
    // ---------------------------------------------------------------- prolongation (cc_prolong_add) and the post-smoothing of F
    __syncthreads();
    if (on) W[w0] = xc;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TAIL_CPT; ++m) if (onF[m]) x[m] = x[m] + W[cF[m]];
    sweepsF(nu2);
#pragma unroll
    for (int m = 0; m < TAIL_CPT; ++m)
        if (onF[m]) {
            const int idx = tid + m * TAIL_NT;
            corF(bF.lo[0] + idx % nxF, bF.lo[1] + (idx / nxF) % nyF, bF.lo[2] + idx / (nxF * nyF), 0) = x[m];
        }
}

// the last two levels of a hierarchy can run as k_abec_tail: F one box of at most 16 cells a side (4096 cells) spanning its domain, C its
// coarsening by 2 with the device bottom solver's conditions; one component, one boundary condition set, b arrays or constants
bool abec_tail_ok(const Geometry& gF, const Layout& lF, const Geometry& gC, const Layout& lC, const AbecCoef& cF, const DomainBC* bcs, int nbc, int ncomp)
{
    // opt-in (IAMRX_MG_TAIL_FUSED = 1): measured on MI355X at 256^3 -- 97.6 us per launch against ~83 us for the 14 launches it replaces (one CU
    // does serially what they spread over the chip; 117 launches fewer per step, step time equal within the noise)
    if (tune("MG_TAIL_FUSED", 0) == 0 || ncomp != 1 || nbc != 1 || cF.sig || cF.tensor || cF.tensor_eta) return false;
    if (!abec_bottom_device_ok(gC, lC, bcs, nbc, ncomp, false) || lF.boxes.size() != 1) return false;
    const BoxD bF = lF.boxes[0], bC = lC.boxes[0];
    for (int d = 0; d < 3; ++d) {
        if (bF.len(d) > TAIL_NF || bF.len(d) != 2 * bC.len(d) || bF.lo[d] != 2 * bC.lo[d]) return false;
        if (bF.lo[d] != gF.domain.lo[d] || bF.hi[d] != gF.domain.hi[d]) return false;
        if (gF.periodic[d]) continue;
        for (int side = 0; side < 2; ++side) {
            const int t = side == 0 ? bcs[0].lo[d] : bcs[0].hi[d];
            if (t != lo_neumann && t != lo_dirichlet && t != lo_reflect_odd) return false;
        }
    }
    if (!cF.b_uniform && cF.b[0]->ncomp != 1) return false;
    return bF.npts() <= (long)TAIL_NT * TAIL_CPT;
}

void abec_tail_solve(const Geometry& gF, const AbecCoef& cF, MultiFab& corF, const MultiFab& resF, const Geometry& gC, const AbecCoef& cC,
                     const DomainBC& bc, bool singular, double eps_rel, int maxiter, int nub, int nuf, int nu1, int nu2, double omega, int* d_iters)
{
    const Layout& lF = *corF.layout;
    if (lF.nlocal() == 0) return;
    TailLevel F, C;
    F.b = lF.boxes[0];
    C.b = coarsen(F.b, 2);
    make_bot_bc(gF, F.b, bc, nullptr, F.bc, F.gb);
    make_bot_bc(gC, C.b, bc, nullptr, C.bc, C.gb);
    F.dhx = cF.beta / (gF.dx[0] * gF.dx[0]); F.dhy = cF.beta / (gF.dx[1] * gF.dx[1]); F.dhz = cF.beta / (gF.dx[2] * gF.dx[2]);
    C.dhx = cC.beta / (gC.dx[0] * gC.dx[0]); C.dhy = cC.beta / (gC.dx[1] * gC.dx[1]); C.dhz = cC.beta / (gC.dx[2] * gC.dx[2]);
    const bool uni = cF.b_uniform && cC.b_uniform && abec_sig_on();
    TailTabs tf{corF.d_tab, resF.d_tab, cF.a ? cF.a->d_tab : nullptr, cF.b[0]->d_tab, cF.b[1]->d_tab, cF.b[2]->d_tab};
    TailTabs tc{nullptr, nullptr, cC.a ? cC.a->d_tab : nullptr, cC.b[0]->d_tab, cC.b[1]->d_tab, cC.b[2]->d_tab};
    BUni bu;
    for (int d = 0; d < 3; ++d) bu.v[d] = cF.bu[d];
    hipLaunchKernelGGL(k_abec_tail, dim3(1), dim3(TAIL_NT), 0, Context::get().stream, F, C, tf, tc, cF.alpha, uni ? 1 : 0, bu, nu1, nu2, omega,
                       singular ? 1 : 0, eps_rel, maxiter, nub, nuf, d_iters);
}

// ---------------------------------------------------------------------------- domain BC ghost fill
struct BndryDesc { int fab; BoxD region; int dir, side; };

struct BcParams { int bct[6]; double c[6][5]; };   // per face (2*d + side): LinOpBC type; Dirichlet weights c[0..3] and NX
// one set for all components (nset == 1) or one per component (the tensor solves: Diffusion.cpp:724-731), in ONE launch
struct BcParamsN { BcParams p[3]; int nset; };

__global__ void __launch_bounds__(256) k_abec_bc(const BndryDesc* __restrict__ descs, const FabD* __restrict__ phit,
                                                 const FabD* __restrict__ bcvt, int ncomp, int comp0, BcParamsN PN, int inhomog)
{
    const BndryDesc bd = descs[blockIdx.y];
    const FabD phi = phit[bd.fab];
    const int nx = bd.region.len(0), ny = bd.region.len(1);
    const long npts = bd.region.npts();
    const int d = bd.dir, s = 1 - 2 * bd.side;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        int idx[3];
        idx[0] = bd.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        idx[1] = bd.region.lo[1] + (int)(r % ny);
        idx[2] = bd.region.lo[2] + (int)(r / ny);
        for (int n = comp0; n < comp0 + ncomp; ++n) {
            const BcParams& P = PN.p[PN.nset == 1 ? 0 : n - comp0];
            const int bct = P.bct[2 * d + bd.side];
            if (bct != lo_neumann && bct != lo_dirichlet && bct != lo_reflect_odd) continue;
            const double* c = P.c[2 * d + bd.side];   // c[0..3] weights, c[4] = NX
            const int NX = (int)c[4];
            double v;
            int m[3] = {idx[0], idx[1], idx[2]};
            if (bct == lo_neumann) { m[d] += s; v = phi(m[0], m[1], m[2], n); }
            else if (bct == lo_reflect_odd) { m[d] += s; v = -phi(m[0], m[1], m[2], n); }
            else {
                const double bv = (inhomog && bcvt) ? bcvt[bd.fab](idx[0], idx[1], idx[2], n) : 0.0;
                if (NX < 2) v = bv;
                else {
                    double tmp = 0.0;
                    for (int q2 = 1; q2 < NX; ++q2) { m[d] = idx[d] + q2 * s; tmp += phi(m[0], m[1], m[2], n) * c[q2]; }
                    v = tmp + bv * c[0];
                }
            }
            phi(idx[0], idx[1], idx[2], n) = v;
        }
    }
}

// The ghost slabs of the box faces on the non-periodic domain boundary depend on the layout and the domain only: built and
// uploaded once per (layout, domain) and kept (a level's multigrid calls this for every colour pass of every smooth).
namespace {
struct BcDescCache { BndryDesc* d = nullptr; int n = 0; long maxpts = 0; };
std::map<std::array<long, 8>, BcDescCache>& bc_desc_cache()
{
    // never destroyed; key[0] = layout id, entries leave with their layout
    static auto* m = [] {
        auto* c = new std::map<std::array<long, 8>, BcDescCache>();
        register_layout_evictor([c](uint64_t lid) {
            for (auto it = c->begin(); it != c->end();) {
                if ((uint64_t)it->first[0] == lid) { if (it->second.d) (void)hipFree(it->second.d); it = c->erase(it); } else ++it;
            }
        });
        return c;
    }();
    return *m;
}
}  // namespace

static void abec_apply_domain_bc_sets(const Geometry& g, MultiFab& phi, const DomainBC* bcs, int nset, bool inhomog, const MultiFab* bcval, int comp0, int ncomp);

void abec_apply_domain_bc(const Geometry& g, MultiFab& phi, const DomainBC& bc, bool inhomog, const MultiFab* bcval, int comp0, int ncomp)
{
    abec_apply_domain_bc_sets(g, phi, &bc, 1, inhomog, bcval, comp0, ncomp);
}

// one boundary-condition set per component (at most 3), all components in one launch
void abec_apply_domain_bc_percomp(const Geometry& g, MultiFab& phi, const DomainBC* bcs, int ncomp, bool inhomog, const MultiFab* bcval)
{
    IAMRX_ASSERT(ncomp >= 1 && ncomp <= 3 && ncomp <= phi.ncomp);
    abec_apply_domain_bc_sets(g, phi, bcs, ncomp, inhomog, bcval, 0, ncomp);
}

static void abec_apply_domain_bc_sets(const Geometry& g, MultiFab& phi, const DomainBC* bcs, int nset, bool inhomog, const MultiFab* bcval, int comp0, int ncomp)
{
    if (ncomp < 0) ncomp = phi.ncomp - comp0;
    bool any = false;
    for (int d = 0; d < 3; ++d) if (!g.periodic[d]) any = true;
    if (!any || phi.nlocal() == 0) return;
    auto& ctx = Context::get();
    const std::array<long, 8> key = {(long)phi.layout->id, g.domain.lo[0], g.domain.lo[1], g.domain.lo[2], g.domain.hi[0], g.domain.hi[1], g.domain.hi[2],
                                     (long)(g.periodic[0] | (g.periodic[1] << 1) | (g.periodic[2] << 2))};
    auto& cache = bc_desc_cache();
    auto it = cache.find(key);
    if (it == cache.end()) {
        std::vector<BndryDesc> descs;
        BcDescCache e;
        for (int li = 0; li < phi.nlocal(); ++li) {
            const BoxD vb = phi.layout->lbox(li);
            for (int d = 0; d < 3; ++d) {
                if (g.periodic[d]) continue;
                for (int side = 0; side < 2; ++side) {
                    if (side == 0 && vb.lo[d] != g.domain.lo[d]) continue;
                    if (side == 1 && vb.hi[d] != g.domain.hi[d]) continue;
                    BndryDesc bd; bd.fab = li; bd.dir = d; bd.side = side; bd.region = vb;
                    bd.region.lo[d] = bd.region.hi[d] = side == 0 ? vb.lo[d] - 1 : vb.hi[d] + 1;
                    descs.push_back(bd);
                    e.maxpts = std::max(e.maxpts, bd.region.npts());
                }
            }
        }
        e.n = (int)descs.size();
        if (e.n > 0) {
            IAMRX_HIP_CHECK(hipMalloc(&e.d, descs.size() * sizeof(BndryDesc)));
            IAMRX_HIP_CHECK(hipMemcpy(e.d, descs.data(), descs.size() * sizeof(BndryDesc), hipMemcpyHostToDevice));
        }
        it = cache.emplace(key, e).first;
    }
    const BcDescCache& e = it->second;
    if (e.n == 0) return;
    BcParamsN PN;
    PN.nset = nset;
    for (int m = 0; m < 3; ++m) {
        const DomainBC& bc = bcs[m < nset ? m : 0];
        BcParams& P = PN.p[m];
        for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) {
            P.bct[2 * d + side] = side == 0 ? bc.lo[d] : bc.hi[d];
            double c[4]; int NX; dirichlet_coefs(g.domain.len(d), bc.maxorder, c, NX);
            for (int q = 0; q < 4; ++q) P.c[2 * d + side][q] = c[q];
            P.c[2 * d + side][4] = NX;
        }
    }
    long nb = (e.maxpts + 255) / 256; if (nb > 128) nb = 128; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_abec_bc, dim3((unsigned)nb, (unsigned)e.n), dim3(256), 0, ctx.stream,
                       (const BndryDesc*)e.d, phi.d_tab, bcval ? bcval->d_tab : nullptr, ncomp, comp0, PN, inhomog ? 1 : 0);
}

// ---------------------------------------------------------------------------- coarse/fine faces
CfTab cf_make_tab(const double loc[3], const double dx[3], int maxorder)
{
    CfTab t;
    t.maxorder = maxorder < 2 ? 2 : (maxorder > 4 ? 4 : maxorder);
    for (int d = 0; d < 3; ++d)
        for (int q = 0; q < 3; ++q) {
            const int NX = q + 2;
            const double x[4] = {-loc[d] / dx[d], 0.5, 1.5, 2.5};
            for (int m = 0; m < 4; ++m) t.c[d][q][m] = 0.0;
            poly_interp_coeff(-0.5, x, NX, t.c[d][q]);
        }
    return t;
}

void cf_build_mask(const Geometry& g, MultiFab& cfm)
{
    cfm.setVal(1.0);
    cfm.setVal(0.0, 0, 1, 0);
    cfm.FillBoundary(g);                 // ghost cells covered by another box of the level or by a periodic image become 0
    if (cfm.nlocal() == 0) return;
    const FabD* ct = cfm.d_tab;
    const BoxD dom = g.domain;
    const int p0 = g.periodic[0], p1 = g.periodic[1], p2 = g.periodic[2];
    for_each(*cfm.layout, cell_type(), cfm.ngrow, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const bool out = (!p0 && (i < dom.lo[0] || i > dom.hi[0])) || (!p1 && (j < dom.lo[1] || j > dom.hi[1])) || (!p2 && (k < dom.lo[2] || k > dom.hi[2]));
        if (out) ct[f](i, j, k) = 2.0;
    });
}

// grid: x = points of one ghost slab, y = box, z = the six slabs of the one-cell shell around the box (slab 2 d + side: the ghost layer
// of direction d, transverse ranges grown by one so that edges and corners are covered; a cell on an edge belongs to two or three
// slabs and receives the same value from each).  The shell is 6 n^2 points, the grown box the kernel used to scan (n + 2)^3.
// list (a level of many unequal boxes, cf_fill_ghosts): entry b = (box, slab, first point of the workgroup) instead of a grid sized for the
// largest slab of the largest box
__global__ void __launch_bounds__(256) k_cf_fill(const BoxD* __restrict__ boxes, const FabD* __restrict__ phit,
    const FabD* __restrict__ bcvt, const FabD* __restrict__ cfmt, int ncomp, CfTab tab, int inhomog, int edges, const int4* __restrict__ list)
{
    int fab = blockIdx.y, slab = blockIdx.z;
    long q0 = (long)blockIdx.x * 256;
    if (list) { const int4 e = list[blockIdx.x]; fab = e.x; slab = e.y; q0 = e.z; }
    const BoxD vb = boxes[fab];
    const int sd = slab >> 1, side = slab & 1;
    const int da = sd == 0 ? 1 : 0, db = sd == 2 ? 1 : 2;                // transverse directions, da < db
    const int na = vb.len(da) + 2, nb = vb.len(db) + 2;
    const long q = q0 + threadIdx.x;
    if (q >= (long)na * nb) return;
    int idx3[3];
    idx3[sd] = side == 0 ? vb.lo[sd] - 1 : vb.hi[sd] + 1;
    idx3[da] = vb.lo[da] - 1 + (int)(q % na);
    idx3[db] = vb.lo[db] - 1 + (int)(q / na);
    const int i = idx3[0], j = idx3[1], k = idx3[2];
    const FabD phi = phit[fab], cfm = cfmt[fab];
    {
        const int idx[3] = {i, j, k};
        int d = -1, nout = 0, s = 0;
        for (int e = 0; e < 3; ++e) {
            if (idx[e] < vb.lo[e]) { ++nout; d = e; s = 1; }
            else if (idx[e] > vb.hi[e]) { ++nout; d = e; s = -1; }
        }
        if (cfm(i, j, k) != 1.0) return;
        if (nout >= 2) {
            // tensor operator: edge / corner coarse-fine ghost cells (read by the cross terms only) hold the coarse data interpolated to
            // the cell centre (cf_interp_edges), frozen during the solve; zero in the correction form
            if (edges) for (int n = 0; n < ncomp; ++n) phi(i, j, k, n) = (inhomog && bcvt) ? (double)bcvt[fab](i, j, k, n) : 0.0;
            return;
        }
        if (nout != 1) return;                               // face-adjacent coarse/fine ghost cells
        const int NX = min(vb.len(d) + 1, tab.maxorder);
        const double* c = tab.c[d][NX - 2];
        for (int n = 0; n < ncomp; ++n) {
            double v = (inhomog && bcvt) ? bcvt[fab](i, j, k, n) * c[0] : 0.0;
            int m[3] = {i, j, k};
            for (int q = 1; q < NX; ++q) { m[d] = idx[d] + q * s; v += phi(m[0], m[1], m[2], n) * c[q]; }
            phi(i, j, k, n) = v;
        }
    }
}

void cf_fill_ghosts(MultiFab& phi, const MultiFab& cfm, const CfTab& tab, bool inhomog, const MultiFab* bcval, bool edges)
{
    if (phi.nlocal() == 0) return;
    const Layout& l = *phi.layout;
    int m[3] = {l.max_len[0] + 2, l.max_len[1] + 2, l.max_len[2] + 2};
    std::sort(m, m + 3);
    const long maxpts = (long)m[1] * m[2];                    // the largest slab
    if (l.nlocal() >= 4 && tune("TILE_LISTS", 1) != 0) {
        int n = 0;
        const int4* lst = layout_int4_list(l, {2, 0, 0, 0, 0}, [&](std::vector<int4>& h) {
            for (int f = 0; f < l.nlocal(); ++f) {
                const BoxD b = l.lbox(f);
                for (int sl = 0; sl < 6; ++sl) {
                    const int sd = sl >> 1, da = sd == 0 ? 1 : 0, db = sd == 2 ? 1 : 2;
                    const long np = (long)(b.len(da) + 2) * (b.len(db) + 2);
                    for (long q0 = 0; q0 < np; q0 += 256) h.push_back(make_int4(f, sl, (int)q0, 0));
                }
            }
        }, &n);
        if (lst && 4L * n <= 3L * ((maxpts + 255) / 256) * l.nlocal() * 6) {
            hipLaunchKernelGGL(k_cf_fill, dim3((unsigned)n), dim3(256), 0, Context::get().stream,
                               l.d_boxes, phi.d_tab, bcval ? bcval->d_tab : nullptr, cfm.d_tab, phi.ncomp, tab, inhomog ? 1 : 0, edges ? 1 : 0, lst);
            return;
        }
    }
    hipLaunchKernelGGL(k_cf_fill, dim3((unsigned)((maxpts + 255) / 256), (unsigned)l.nlocal(), 6u), dim3(256), 0, Context::get().stream,
                       l.d_boxes, phi.d_tab, bcval ? bcval->d_tab : nullptr, cfm.d_tab, phi.ncomp, tab, inhomog ? 1 : 0, edges ? 1 : 0, (const int4*)nullptr);
}

// bcval(edge / corner coarse-fine ghost cells) = the coarse data of cpatch interpolated to the cell centre, quadratically in every
// direction: centred stencil (c-1, c, c+1) in the directions in which the cell lies inside the box's index range, one-sided towards
// the box (c, c+s, c+2s) in the directions in which it lies outside (third order: the cross terms difference these values over h)
void cf_interp_edges(MultiFab& bcval, const MultiFab& cpatch, const MultiFab& cfm, int ratio, const Geometry& cgeom)
{
    if (bcval.nlocal() == 0) return;
    IAMRX_ASSERT(cpatch.ngrow >= 1 && bcval.ngrow >= 1 && cfm.ngrow >= 1);
    const FabD *bt = bcval.d_tab, *ct = cpatch.d_tab, *mt = cfm.d_tab;
    const BoxD* boxes = bcval.layout->d_boxes;
    const int nc = bcval.ncomp, r = ratio;
    const BoxD cdom = cgeom.domain;
    const int cper[3] = {cgeom.periodic[0], cgeom.periodic[1], cgeom.periodic[2]};
    const int cp0 = cper[0], cp1 = cper[1], cp2 = cper[2];
    for_each(*bcval.layout, cell_type(), 1, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const BoxD vb = boxes[f];
        const int q[3] = {i, j, k};
        const int per[3] = {cp0, cp1, cp2};
        int nout = 0;
        for (int e = 0; e < 3; ++e) if (q[e] < vb.lo[e] || q[e] > vb.hi[e]) ++nout;
        if (nout < 2 || mt[f](i, j, k) != 1.0) return;
        int c[3], o[3][3];
        double w[3][3];
        for (int e = 0; e < 3; ++e) {
            c[e] = q[e] >= 0 ? q[e] / r : -((-q[e] + r - 1) / r);
            const double off = (q[e] - c[e] * r + 0.5) / r - 0.5;
            if (q[e] < vb.lo[e] || q[e] > vb.hi[e]) {
                const int s = q[e] < vb.lo[e] ? 1 : -1;
                const double u = off * s;
                o[e][0] = 0; o[e][1] = s; o[e][2] = 2 * s;
                w[e][0] = 0.5 * (u - 1.0) * (u - 2.0); w[e][1] = -u * (u - 2.0); w[e][2] = 0.5 * u * (u - 1.0);
            } else if (!per[e] && c[e] - 1 < cdom.lo[e]) {        // next to a wall: one-sided, no coarse cell outside the physical domain
                o[e][0] = 0; o[e][1] = 1; o[e][2] = 2;
                w[e][0] = 0.5 * (off - 1.0) * (off - 2.0); w[e][1] = -off * (off - 2.0); w[e][2] = 0.5 * off * (off - 1.0);
            } else if (!per[e] && c[e] + 1 > cdom.hi[e]) {
                o[e][0] = -2; o[e][1] = -1; o[e][2] = 0;
                w[e][0] = 0.5 * off * (off + 1.0); w[e][1] = -off * (off + 2.0); w[e][2] = 0.5 * (off + 1.0) * (off + 2.0);
            } else {
                o[e][0] = -1; o[e][1] = 0; o[e][2] = 1;
                w[e][0] = 0.5 * off * (off - 1.0); w[e][1] = 1.0 - off * off; w[e][2] = 0.5 * off * (off + 1.0);
            }
        }
        const FabD cp = ct[f];
        for (int n = 0; n < nc; ++n) {
            double v = 0.0;
            for (int cz = 0; cz < 3; ++cz) for (int cy = 0; cy < 3; ++cy) for (int cx = 0; cx < 3; ++cx)
                v += w[0][cx] * w[1][cy] * w[2][cz] * cp(c[0] + o[0][cx], c[1] + o[1][cy], c[2] + o[2][cz], n);
            bt[f](i, j, k, n) = v;
        }
    });
}

// interpbndrydata_{x,y,z}_o3: value at the fine ghost cell from the coarse cell under it and its tangential neighbours; a
// neighbour takes part only if the ghost cell `ratio` cells further along is a coarse/fine ghost cell too (mask 1)
__global__ void __launch_bounds__(256) k_cf_interp(Tiling t, const BoxD* __restrict__ boxes, const FabD* __restrict__ bcvt,
    const FabD* __restrict__ cpt, const FabD* __restrict__ cfmt, int ncomp, int r)
{
    const int fab = blockIdx.y;
    const BoxD vb = boxes[fab];
    BoxD gb = vb;
    for (int d = 0; d < 3; ++d) { gb.lo[d] -= 1; gb.hi[d] += 1; }
    int i, j, k0, k1;
    if (!tile_ijk(t, gb, i, j, k0, k1)) return;
    const FabD bcv = bcvt[fab], cp = cpt[fab], cfm = cfmt[fab];
    for (int k = k0; k <= k1; ++k) {
        const int q[3] = {i, j, k};
        int d = -1, nout = 0;
        for (int e = 0; e < 3; ++e) if (q[e] < vb.lo[e] || q[e] > vb.hi[e]) { ++nout; d = e; }
        if (nout != 1 || cfm(i, j, k) != 1.0) continue;
        const int t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
        int c[3];
        for (int e = 0; e < 3; ++e) c[e] = q[e] >= 0 ? q[e] / r : -((-q[e] + r - 1) / r);
        auto msk = [&](int o1, int o2) { int a[3] = {i, j, k}; a[t1] += o1 * r; a[t2] += o2 * r; return cfm(a[0], a[1], a[2]) == 1.0; };
        const bool m1m = msk(-1, 0), m1p = msk(1, 0), m2m = msk(0, -1), m2p = msk(0, 1);
        const bool diag = msk(-1, -1) && msk(1, -1) && msk(-1, 1) && msk(1, 1);
        const int l1 = m1m ? -1 : 0, h1 = m1p ? 1 : 0, l2 = m2m ? -1 : 0, h2 = m2p ? 1 : 0;
        const double f1 = (h1 == l1 + 1) ? 1.0 : 0.5, f2 = (h2 == l2 + 1) ? 1.0 : 0.5;
        const double y1 = -0.5 + (q[t1] - c[t1] * r + 0.5) / r, y2 = -0.5 + (q[t2] - c[t2] * r + 0.5) / r;
        for (int n = 0; n < ncomp; ++n) {
            auto CC = [&](int o1, int o2) { int z[3] = {c[0], c[1], c[2]}; z[t1] += o1; z[t2] += o2; return (double)cp(z[0], z[1], z[2], n); };
            const double d1 = f1 * (CC(h1, 0) - CC(l1, 0));
            const double d11 = (h1 == l1 + 2) ? 0.5 * (CC(1, 0) - 2.0 * CC(0, 0) + CC(-1, 0)) : 0.0;
            const double d2 = f2 * (CC(0, h2) - CC(0, l2));
            const double d22 = (h2 == l2 + 2) ? 0.5 * (CC(0, 1) - 2.0 * CC(0, 0) + CC(0, -1)) : 0.0;
            const double d12 = diag ? 0.25 * (CC(1, 1) - CC(-1, 1) + CC(-1, -1) - CC(1, -1)) : 0.0;
            bcv(i, j, k, n) = CC(0, 0) + y1 * d1 + (y1 * y1) * d11 + y2 * d2 + (y2 * y2) * d22 + y1 * y2 * d12;
        }
    }
}

void cf_interp_bndry(MultiFab& bcval, const MultiFab& cpatch, const MultiFab& cfm, int ratio)
{
    if (bcval.nlocal() == 0) return;
    IAMRX_ASSERT(cfm.ngrow >= 1 + ratio - 1 + 1 - 1 && cfm.ngrow >= 2 && cpatch.ngrow >= 1 && bcval.ngrow >= 1 && ratio == 2);
    Tiling t = level_tiling(*bcval.layout, cell_type(), 1, 4);
    hipLaunchKernelGGL(k_cf_interp, t.grid(), Tiling::block(), 0, Context::get().stream, t, bcval.layout->d_boxes, bcval.d_tab,
                       cpatch.d_tab, cfm.d_tab, bcval.ncomp, ratio);
}

// ---------------------------------------------------------------------------- transfers
__global__ void __launch_bounds__(256) k_cc_restrict(Tiling t, const BoxD* __restrict__ cboxes, const FabD* __restrict__ ct,
                                                     const FabD* __restrict__ ft, int ncomp)
{
    const int fab = blockIdx.y;
    int i, j, k0, k1;
    if (!tile_ijk(t, cboxes[fab], i, j, k0, k1)) return;
    const FabD c = ct[fab], f = ft[fab];
    for (int n = 0; n < ncomp; ++n)
        for (int k = k0; k <= k1; ++k) {
            double s = 0.0;
            for (int kr = 0; kr < 2; ++kr)
                for (int jr = 0; jr < 2; ++jr) {
                    s += f(2 * i, 2 * j + jr, 2 * k + kr, n);
                    s += f(2 * i + 1, 2 * j + jr, 2 * k + kr, n);
                }
            c(i, j, k, n) = 0.125 * s;
        }
}

void cc_restrict(MultiFab& crse, const MultiFab& fine)
{
    if (crse.nlocal() == 0) return;
    Tiling t = level_tiling(*crse.layout, cell_type(), 0, 4);
    hipLaunchKernelGGL(k_cc_restrict, t.grid(), Tiling::block(), 0, Context::get().stream, t, crse.layout->d_boxes, crse.d_tab, fine.d_tab, crse.ncomp);
}

void cc_prolong_add(MultiFab& fine, const MultiFab& crse)
{
    if (fine.nlocal() == 0) return;
    const FabD *ft = fine.d_tab, *ct = crse.d_tab;
    const int nc = fine.ncomp;
    for_each_2ph(*fine.layout, cell_type(), 0, nc, Context::get().stream,
        [=] __device__(int i, int j, int k, int f, int n) { return ft[f](i, j, k, n) + ct[f](i >> 1, j >> 1, k >> 1, n); },
        [=] __device__(int i, int j, int k, int f, int n, double v) { ft[f](i, j, k, n) = v; });
}

void face_avgdown(MultiFab& crse, const MultiFab& fine, int dir)
{
    if (crse.nlocal() == 0) return;
    const FabD *ft = fine.d_tab, *ct = crse.d_tab;
    const int nc = crse.ncomp;
    const int d1 = (dir + 1) % 3, d2 = (dir + 2) % 3;
    const int da = d1 < d2 ? d1 : d2, db = d1 < d2 ? d2 : d1;
    for_each(*crse.layout, face_type(dir), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD fa = ft[f], ca = ct[f];
        const int c[3] = {i, j, k};
        for (int n = 0; n < nc; ++n) {
            double s = 0.0;
            for (int rb = 0; rb < 2; ++rb)
                for (int ra = 0; ra < 2; ++ra) {
                    int q[3];
                    q[dir] = 2 * c[dir]; q[da] = 2 * c[da] + ra; q[db] = 2 * c[db] + rb;
                    s += fa(q[0], q[1], q[2], n);
                }
            ca(i, j, k, n) = s * 0.25;
        }
    });
}

// ---------------------------------------------------------------------------- fluxes / MAC helpers
void abec_flux(const Geometry& g, const AbecCoef& c, const MultiFab& phi, MultiFab* const flux[3], MultiFab* const add_to[3])
{
    if (phi.nlocal() == 0) return;
    const FabD* pt = phi.d_tab;
    const int nc = phi.ncomp;
    for (int d = 0; d < 3; ++d) {
        const double fac = c.beta / g.dx[d];
        const FabD* bt = c.b[d]->d_tab;
        const int bnc = c.b[d]->ncomp;
        const FabD* ft = flux && flux[d] ? flux[d]->d_tab : nullptr;
        const FabD* ut = add_to && add_to[d] ? add_to[d]->d_tab : nullptr;
        for_each(*phi.layout, face_type(d), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
            const FabD p = pt[f], b = bt[f];
            int m[3] = {i, j, k}; m[d] -= 1;
            for (int n = 0; n < nc; ++n) {
                const double fl = -fac * b(i, j, k, bnc == 1 ? 0 : n) * (p(i, j, k, n) - p(m[0], m[1], m[2], n));
                if (ft) ft[f](i, j, k, n) = fl;
                if (ut) ut[f](i, j, k, n) += fl;
            }
        });
    }
}

void mac_divergence(const Geometry& g, MultiFab& div, const MultiFab* const umac[3])
{
    if (div.nlocal() == 0) return;
    const FabD *dt = div.d_tab, *ut = umac[0]->d_tab, *vt = umac[1]->d_tab, *wt = umac[2]->d_tab;
    const double dx = g.dx[0], dy = g.dx[1], dz = g.dx[2];
    for_each(*div.layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD u = ut[f], v = vt[f], w = wt[f];
        dt[f](i, j, k, 0) = (u(i + 1, j, k) - u(i, j, k)) / dx + (v(i, j + 1, k) - v(i, j, k)) / dy + (w(i, j, k + 1) - w(i, j, k)) / dz;
    });
}

void mac_rhs(const Geometry& g, MultiFab& rhs, const MultiFab* const umac[3], const MultiFab* S)
{
    if (rhs.nlocal() == 0) return;
    const FabD *rt = rhs.d_tab, *ut = umac[0]->d_tab, *vt = umac[1]->d_tab, *wt = umac[2]->d_tab;
    const FabD* st = S ? S->d_tab : nullptr;
    const double dx = g.dx[0], dy = g.dx[1], dz = g.dx[2];
    for_each(*rhs.layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD u = ut[f], v = vt[f], w = wt[f];
        const double div = (u(i + 1, j, k) - u(i, j, k)) / dx + (v(i, j + 1, k) - v(i, j, k)) / dy + (w(i, j, k + 1) - w(i, j, k)) / dz;
        double r = -div;
        if (st) r += st[f](i, j, k, 0);
        rt[f](i, j, k, 0) = r;
    });
}

void mac_bcoef(MultiFab* const b[3], const MultiFab& rho, int rho_comp, double scale)
{
    if (rho.nlocal() == 0) return;
    const FabD* rt = rho.d_tab;
    for (int d = 0; d < 3; ++d) {
        const FabD* bt = b[d]->d_tab;
        for_each(*rho.layout, face_type(d), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
            const FabD r = rt[f];
            int m[3] = {i, j, k}; m[d] -= 1;
            const double rf = 0.5 * (r(m[0], m[1], m[2], rho_comp) + r(i, j, k, rho_comp));
            bt[f](i, j, k, 0) = scale / rf;
        });
    }
}

// ---------------------------------------------------------------------------- uniform coefficient detection
bool mf_uniform_value(const MultiFab& m, double* v)
{
    auto& ctx = Context::get();
    static double* d_out = nullptr;                      // [0] flag (1: some entry differs from the first one), [1] the first entry
    if (!d_out) IAMRX_HIP_CHECK(hipMalloc(&d_out, 2 * sizeof(double)));
    double h[3] = {0.0, -1.e300, -1.e300};               // flag, max of the reference values, max of their negatives
    if (m.nlocal() > 0) {
        IAMRX_HIP_CHECK(hipMemsetAsync(d_out, 0, 2 * sizeof(double), ctx.stream));
        const FabD* mt = m.d_tab;
        const BoxD b0 = m.layout->lbox(0);
        double* out = d_out;
        for_each(*m.layout, m.type, 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
            const double ref = mt[0](b0.lo[0], b0.lo[1], b0.lo[2], 0);
            if (f == 0 && i == b0.lo[0] && j == b0.lo[1] && k == b0.lo[2]) out[1] = ref;
            if (!(mt[f](i, j, k, 0) == ref)) out[0] = 1.0;
        });
        double r[2];
        IAMRX_HIP_CHECK(hipMemcpyAsync(r, d_out, sizeof(r), hipMemcpyDeviceToHost, ctx.stream));
        ctx.sync();
        h[0] = r[0]; h[1] = r[1]; h[2] = -r[1];
    }
    if (!m.layout->replicated && ctx.comm->nranks > 1) ctx.comm->allreduce(h, 3, ReduceOp::Max);
    *v = h[1];
    return h[0] == 0.0 && h[1] == -h[2] && h[1] > -1.e299;
}

}  // namespace iamrx
