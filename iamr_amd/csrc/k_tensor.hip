// iamr_amd/csrc/k_tensor.hip -- cross-derivative terms of the viscous stress tensor
// div( eta (grad u + grad u^T) - 2/3 eta (div u) I ) on top of the 3-component ABec operator
// (b_d(comp) = eta * (comp == d ? 4/3 : 1)).  Role: AMReX MLTensorOp (mltensor_cross_terms_f{x,y,z},
// mltensor_cross_terms) as used at reference Source/Diffusion.cpp:715-768 (explicit apply) and
// :858-923 (implicit solve); SURVEY a10, a11.
#include "kernels.h"
#include "launch.h"
#include <cstring>
#include <type_traits>
#include <vector>
#include <cstdlib>
#include <algorithm>

namespace iamrx {

// cross flux on the d-face (i,j,k) for component n.  eta_n = b_d(comp d) * 3/4 (normal), eta_t = b_d(comp != d)
// ETA1: eta holds the 1-component face viscosity (b_d(comp) is formed here exactly as tensor_bcoef stores it)
// the face viscosity of a constant-viscosity operator (AbecCoef::b_uniform) in the place of its array
struct ConstEta {
    double v;
    __device__ __forceinline__ double operator()(int, int, int, int) const { return v; }
};
template <int D, bool ETA1, class VA, class EA>
__device__ __forceinline__ void cross_flux(const VA& v, const EA& eta, int i, int j, int k, double dxi, double dyi, double dzi, double f[3])
{
    constexpr double twoThirds = 2.0 / 3.0;
    const double e1 = ETA1 ? eta(i, j, k, 0) : 0.0;
    const double bn = ETA1 ? e1 * (4.0 / 3.0) : eta(i, j, k, D);        // normal component coefficient
    const double bt = ETA1 ? e1 * 1.0 : eta(i, j, k, D == 0 ? 1 : 0);   // tangential
    if (D == 0) {
        const double dudy = (v(i, j + 1, k, 0) + v(i - 1, j + 1, k, 0) - v(i, j - 1, k, 0) - v(i - 1, j - 1, k, 0)) * (0.25 * dyi);
        const double dvdy = (v(i, j + 1, k, 1) + v(i - 1, j + 1, k, 1) - v(i, j - 1, k, 1) - v(i - 1, j - 1, k, 1)) * (0.25 * dyi);
        const double dudz = (v(i, j, k + 1, 0) + v(i - 1, j, k + 1, 0) - v(i, j, k - 1, 0) - v(i - 1, j, k - 1, 0)) * (0.25 * dzi);
        const double dwdz = (v(i, j, k + 1, 2) + v(i - 1, j, k + 1, 2) - v(i, j, k - 1, 2) - v(i - 1, j, k - 1, 2)) * (0.25 * dzi);
        const double divu = dvdy + dwdz;
        const double xif = 0.0;
        const double mun = 0.75 * (bn - xif), mut = bt;
        f[0] = -mun * (-twoThirds * divu) - xif * divu;
        f[1] = -mut * dudy;
        f[2] = -mut * dudz;
    } else if (D == 1) {
        const double dudx = (v(i + 1, j, k, 0) + v(i + 1, j - 1, k, 0) - v(i - 1, j, k, 0) - v(i - 1, j - 1, k, 0)) * (0.25 * dxi);
        const double dvdx = (v(i + 1, j, k, 1) + v(i + 1, j - 1, k, 1) - v(i - 1, j, k, 1) - v(i - 1, j - 1, k, 1)) * (0.25 * dxi);
        const double dvdz = (v(i, j, k + 1, 1) + v(i, j - 1, k + 1, 1) - v(i, j, k - 1, 1) - v(i, j - 1, k - 1, 1)) * (0.25 * dzi);
        const double dwdz = (v(i, j, k + 1, 2) + v(i, j - 1, k + 1, 2) - v(i, j, k - 1, 2) - v(i, j - 1, k - 1, 2)) * (0.25 * dzi);
        const double divu = dudx + dwdz;
        const double xif = 0.0;
        const double mun = 0.75 * (bn - xif), mut = bt;
        f[0] = -mut * dvdx;
        f[1] = -mun * (-twoThirds * divu) - xif * divu;
        f[2] = -mut * dvdz;
    } else {
        const double dudx = (v(i + 1, j, k, 0) + v(i + 1, j, k - 1, 0) - v(i - 1, j, k, 0) - v(i - 1, j, k - 1, 0)) * (0.25 * dxi);
        const double dwdx = (v(i + 1, j, k, 2) + v(i + 1, j, k - 1, 2) - v(i - 1, j, k, 2) - v(i - 1, j, k - 1, 2)) * (0.25 * dxi);
        const double dvdy = (v(i, j + 1, k, 1) + v(i, j + 1, k - 1, 1) - v(i, j - 1, k, 1) - v(i, j - 1, k - 1, 1)) * (0.25 * dyi);
        const double dwdy = (v(i, j + 1, k, 2) + v(i, j + 1, k - 1, 2) - v(i, j - 1, k, 2) - v(i, j - 1, k - 1, 2)) * (0.25 * dyi);
        const double divu = dudx + dvdy;
        const double xif = 0.0;
        const double mun = 0.75 * (bn - xif), mut = bt;
        f[0] = -mut * dwdx;
        f[1] = -mut * dwdy;
        f[2] = -mun * (-twoThirds * divu) - xif * divu;
    }
}

// max norm of the launch's output as a by-product (see norm_commit, k_abec.hip)
__device__ __forceinline__ void tnorm_commit(double mx, unsigned long long* out)
{
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.0) {
        // the running maximum only grows: a plain read filters out almost every wavefront before the atomic (262 k atomics on one address
        // cost 130 us per 256^3 launch without it)
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mx);
        if (bits > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, bits);
    }
}

template <bool ETA1>
__global__ void __launch_bounds__(256) k_tensor_cross(Tiling t, const BoxD* __restrict__ boxes, const FabD* __restrict__ outt,
    const FabD* __restrict__ vt, const FabD* __restrict__ ext, const FabD* __restrict__ eyt, const FabD* __restrict__ ezt,
    double dxi, double dyi, double dzi, double sbeta, unsigned long long* __restrict__ normout)
{
    const int fab = blockIdx.y;
    int i, j, k0, k1;
    double mx = 0.0;
    if (!tile_ijk(t, boxes[fab], i, j, k0, k1)) { if (normout) tnorm_commit(mx, normout); return; }
    const FabD out = outt[fab], v = vt[fab], ex = ext[fab], ey = eyt[fab], ez = ezt[fab];
    double fzl[3];
    cross_flux<2, ETA1>(v, ez, i, j, k0, dxi, dyi, dzi, fzl);
    for (int k = k0; k <= k1; ++k) {
        double fxl[3], fxh[3], fyl[3], fyh[3], fzh[3];
        cross_flux<0, ETA1>(v, ex, i, j, k, dxi, dyi, dzi, fxl);
        cross_flux<0, ETA1>(v, ex, i + 1, j, k, dxi, dyi, dzi, fxh);
        cross_flux<1, ETA1>(v, ey, i, j, k, dxi, dyi, dzi, fyl);
        cross_flux<1, ETA1>(v, ey, i, j + 1, k, dxi, dyi, dzi, fyh);
        cross_flux<2, ETA1>(v, ez, i, j, k + 1, dxi, dyi, dzi, fzh);
        for (int n = 0; n < 3; ++n) {
            const double o = out(i, j, k, n) + sbeta * (dxi * (fxh[n] - fxl[n]) + dyi * (fyh[n] - fyl[n]) + dzi * (fzh[n] - fzl[n]));
            out(i, j, k, n) = o;
            const double a = fabs(o);
            mx = fmax(mx, a == a ? a : INFINITY);
            fzl[n] = fzh[n];
        }
    }
    if (normout) tnorm_commit(mx, normout);
}

// LDS-staged form of k_tensor_cross for levels with at least one full tile: a workgroup owns a TX x TY column of cells and marches
// through a chunk of planes with a three-plane ring of the 3-component velocity tile (one ghost cell in x and y) in LDS; the five face
// fluxes of a cell read their 80 velocity values from LDS instead of L1 / L2 (k_tensor_cross: 0.62 ms per 256^3 launch, load-issue
// bound).  Same expressions (cross_flux), same results.
template <int TX, int TY>
struct LdsVel {
    static constexpr int W = TX + 2, H = TY + 2, PS = W * H;
    const double *pm, *p0, *pp;      // planes kc-1, kc, kc+1 (3 components each, component stride PS)
    int kc, i0, j0;
    __device__ __forceinline__ double operator()(int i, int j, int k, int n) const
    {
        const int d = k - kc;
        const double* p = d < 0 ? pm : (d > 0 ? pp : p0);
        return p[(i - i0) + W * (j - j0) + PS * n];
    }
};

struct EtaUni { double v[3]; };
// UNI (with ETA1): the three face viscosities are the constants eu (the arrays are not read)
// FUSE (with UNI): the whole tensor residual / apply in this pass -- the 7-point part of the operator (k_abec_residual's expression for
// b_d(comp) = eta_d (comp == d ? 4/3 : 1)) is formed from the velocity planes already in LDS, out = (rhs - y | y) + sbeta div(cross) is
// written once: the three components are neither re-read from HBM nor written and read again in between.
struct FuseArgs { const FabD* rhst; const FabD* at; double alpha, dhx, dhy, dhz; };
template <bool ETA1, int TX, int TY, bool UNI = false, bool FUSE = false>
__global__ void __launch_bounds__(TX * TY) k_tensor_cross_zm(const BoxD* __restrict__ boxes, const FabD* __restrict__ outt,
    const FabD* __restrict__ vt, const FabD* __restrict__ ext, const FabD* __restrict__ eyt, const FabD* __restrict__ ezt,
    double dxi, double dyi, double dzi, double sbeta, unsigned long long* __restrict__ normout, int ntx, int nty, int nkc, int kcs, int xcd_cnt,
    EtaUni eu = EtaUni(), FuseArgs fa = FuseArgs())
{
    constexpr int NT = TX * TY, W = TX + 2, H = TY + 2, PS = W * H;
    __shared__ double V[3][3 * PS];
    const int fab = blockIdx.y;
    const BoxD b = boxes[fab];
    double mx = 0.0;
    int bid = blockIdx.x;
    bool live = true;
    if (xcd_cnt > 0) {
        bid = (bid & 7) * xcd_cnt + (bid >> 3);          // XCD-aware order, see make_tiling
        if (bid >= ntx * nty * nkc) live = false;
    }
    const int tix = bid % ntx, r1 = bid / ntx, tiy = r1 % nty, kci = r1 / nty;
    const int tx0 = b.lo[0] + tix * TX, ty0 = b.lo[1] + tiy * TY, k0 = b.lo[2] + kci * kcs;
    if (tx0 > b.hi[0] || ty0 > b.hi[1] || k0 > b.hi[2]) live = false;
    if (!live) { if (normout) tnorm_commit(mx, normout); return; }
    const int k1 = min(k0 + kcs - 1, b.hi[2]);
    const int tid = threadIdx.x;
    const int i = tx0 + tid % TX, j = ty0 + tid / TX;
    const bool on = i <= b.hi[0] && j <= b.hi[1];
    const FabD out = outt[fab], v = vt[fab];
    using EA = typename std::conditional<UNI, ConstEta, FabD>::type;
    EA ex, ey, ez;
    if constexpr (UNI) { ex.v = eu.v[0]; ey.v = eu.v[1]; ez.v = eu.v[2]; }
    else { ex = ext[fab]; ey = eyt[fab]; ez = ezt[fab]; }
    const int vhx = min(tx0 + TX, b.hi[0] + 1), vhy = min(ty0 + TY, b.hi[1] + 1);      // last staged column / row (ghost included)
    auto stage = [&](int k) {
        double* dst = V[((k % 3) + 3) % 3];
        for (int e = tid; e < PS; e += NT) {
            const int ii = tx0 - 1 + e % W, jj = ty0 - 1 + e / W;
            if (ii <= vhx && jj <= vhy) {
                const long o = v.off(ii, jj, k);
#pragma unroll
                for (int n = 0; n < 3; ++n) dst[e + PS * n] = v.gp()[o + v.cs * n];
            }
        }
    };
    // the plane after the next one travels through registers: fetched (loads issued) before the arithmetic of a plane, written to LDS at
    // the top of the next iteration -- staged directly, every plane waited for its loads between two barriers with three workgroups per CU
    constexpr int NE = (PS + NT - 1) / NT;
    double pf[NE][3];
    long pfo[NE];
#pragma unroll
    for (int s_ = 0; s_ < NE; ++s_) {
        const int e = tid + s_ * NT;
        const int ii = tx0 - 1 + e % W, jj = ty0 - 1 + e / W;
        pfo[s_] = (e < PS && ii <= vhx && jj <= vhy) ? v.off(ii, jj, k0) : -1;      // (>= 0: the cell lies in the array)
    }
    const long vks = (long)v.n[0] * v.n[1];
    auto fetch = [&](int k) {
#pragma unroll
        for (int s_ = 0; s_ < NE; ++s_)
            if (pfo[s_] >= 0) {
                const long o = pfo[s_] + vks * (k - k0);
#pragma unroll
                for (int n = 0; n < 3; ++n) pf[s_][n] = v.gp()[o + v.cs * n];
            }
    };
    auto commit = [&](int k) {
        double* dst = V[((k % 3) + 3) % 3];
#pragma unroll
        for (int s_ = 0; s_ < NE; ++s_)
            if (pfo[s_] >= 0) {
                const int e = tid + s_ * NT;
#pragma unroll
                for (int n = 0; n < 3; ++n) dst[e + PS * n] = pf[s_][n];
            }
    };
    stage(k0 - 1);
    stage(k0);
    fetch(k0 + 1);
    __syncthreads();
    LdsVel<TX, TY> a;
    a.i0 = tx0 - 1; a.j0 = ty0 - 1;
    double fzl[3] = {0., 0., 0.};
    {
        a.kc = k0; a.pm = V[((k0 - 1) % 3 + 3) % 3]; a.p0 = V[(k0 % 3 + 3) % 3]; a.pp = a.p0;
        if (on) cross_flux<2, ETA1>(a, ez, i, j, k0, dxi, dyi, dzi, fzl);
    }
    for (int k = k0; k <= k1; ++k) {
        commit(k + 1);
        if (k < k1) fetch(k + 2);
        __syncthreads();
        a.kc = k; a.pm = V[((k - 1) % 3 + 3) % 3]; a.p0 = V[(k % 3 + 3) % 3]; a.pp = V[((k + 1) % 3 + 3) % 3];
        if (on) {
            double fxl[3], fxh[3], fyl[3], fyh[3], fzh[3];
            cross_flux<0, ETA1>(a, ex, i, j, k, dxi, dyi, dzi, fxl);
            cross_flux<0, ETA1>(a, ex, i + 1, j, k, dxi, dyi, dzi, fxh);
            cross_flux<1, ETA1>(a, ey, i, j, k, dxi, dyi, dzi, fyl);
            cross_flux<1, ETA1>(a, ey, i, j + 1, k, dxi, dyi, dzi, fyh);
            cross_flux<2, ETA1>(a, ez, i, j, k + 1, dxi, dyi, dzi, fzh);
            for (int n = 0; n < 3; ++n) {
                double base;
                if constexpr (FUSE) {
                    const double p0 = a(i, j, k, n);
                    const double bx = eu.v[0] * (n == 0 ? 4.0 / 3.0 : 1.0), by = eu.v[1] * (n == 1 ? 4.0 / 3.0 : 1.0), bz = eu.v[2] * (n == 2 ? 4.0 / 3.0 : 1.0);
                    const double ax = fa.at ? fa.alpha * fa.at[fab](i, j, k, 0) * p0 : 0.0;
                    const double y = ax
                        - fa.dhx * (bx * (a(i + 1, j, k, n) - p0) - bx * (p0 - a(i - 1, j, k, n)))
                        - fa.dhy * (by * (a(i, j + 1, k, n) - p0) - by * (p0 - a(i, j - 1, k, n)))
                        - fa.dhz * (bz * (a(i, j, k + 1, n) - p0) - bz * (p0 - a(i, j, k - 1, n)));
                    base = fa.rhst ? fa.rhst[fab](i, j, k, n) - y : y;
                } else base = out(i, j, k, n);
                const double o = base + sbeta * (dxi * (fxh[n] - fxl[n]) + dyi * (fyh[n] - fyl[n]) + dzi * (fzh[n] - fzl[n]));
                out(i, j, k, n) = o;
                const double ab = fabs(o);
                mx = fmax(mx, ab == ab ? ab : INFINITY);
                fzl[n] = fzh[n];
            }
        }
        __syncthreads();
    }
    if (normout) tnorm_commit(mx, normout);
}

// Cell-centred form of the constant-viscosity tensor residual / apply (round 5).  With one face viscosity per direction (eta_x, eta_y,
// eta_z: AbecCoef::b_uniform) the divergence of the cross fluxes telescopes: the transverse derivative on a face is the mean of two
// central differences, the difference of two opposite faces leaves a four-point mixed second difference at the cell centre,
//   C_ab(f) = f(+a, +b) - f(+a, -b) - f(-a, +b) + f(-a, -b),
// and (cross_flux above, D = 0, 1, 2, summed over the six faces of a cell)
//   div(cross)_u = q_xy (2/3 eta_x - eta_y) C_xy(v) + q_xz (2/3 eta_x - eta_z) C_xz(w)
//   div(cross)_v = q_xy (2/3 eta_y - eta_x) C_xy(u) + q_yz (2/3 eta_y - eta_z) C_yz(w)
//   div(cross)_w = q_xz (2/3 eta_z - eta_x) C_xz(u) + q_yz (2/3 eta_z - eta_y) C_yz(v),      q_ab = 1 / (4 h_a h_b).
// 24 values and ~30 flops per cell instead of 80 values and ~160 flops for the five face fluxes of k_tensor_cross_zm: the same operator
// (MLTensorOp with constant shear viscosity, Source/Diffusion.cpp:715-768, 858-923), another summation order -- 1e-12 relative against
// the oracle's face-flux form (tests/test_gpu_kernel_forms.py), not bit for bit.  The 7-point part keeps k_abec_residual's expression.
// Staging: a TX x TY column of cells marches through a z-chunk with a FOUR-slot ring of the 3-component velocity tile in LDS -- plane
// k + 2 is written behind the arithmetic of plane k into the slot of plane k - 2, which nobody reads any more: one barrier per plane.
struct TensorUniArgs {
    double alpha, dhx, dhy, dhz;      // a-term factor, beta / h_d^2
    double b[3][3];                   // b[n][d] = eta_d (n == d ? 4/3 : 1)
    double cx[3][2];                  // sbeta * q * (2/3 eta - eta) of the two mixed differences of component n (order: see kernel)
};
template <int TX, int TY, bool RES, bool HASA>
__global__ void __launch_bounds__(TX * TY) k_tensor_uni(const BoxD* __restrict__ boxes, const FabD* __restrict__ outt, const FabD* __restrict__ vt,
    const FabD* __restrict__ rhst, const FabD* __restrict__ at, TensorUniArgs p, unsigned long long* __restrict__ normout,
    int ntx, int nty, int nkc, int kcs, int xcd_cnt)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = TX * TY, W = TX + 2, H = TY + 2, PS = W * H, NE = (PS + NT - 1) / NT;
    __shared__ double V[4][3 * PS];
    const int fab = blockIdx.y;
    const BoxD b = boxes[fab];
    double mx = 0.0;
    int bid = blockIdx.x;
    bool live = true;
    if (xcd_cnt > 0) {
        bid = (bid & 7) * xcd_cnt + (bid >> 3);          // XCD-aware order, see make_tiling
        if (bid >= ntx * nty * nkc) live = false;
    }
    const int tix = bid % ntx, r1 = bid / ntx, tiy = r1 % nty, kci = r1 / nty;
    const int tx0 = b.lo[0] + tix * TX, ty0 = b.lo[1] + tiy * TY, k0 = b.lo[2] + kci * kcs;
    if (tx0 > b.hi[0] || ty0 > b.hi[1] || k0 > b.hi[2]) live = false;
    if (!live) { if (normout) tnorm_commit(mx, normout); return; }
    const int k1 = min(k0 + kcs - 1, b.hi[2]);
    const int tid = threadIdx.x;
    const int li = tid % TX, lj = tid / TX;
    const int i = tx0 + li, j = ty0 + lj;
    const bool on = i <= b.hi[0] && j <= b.hi[1];
    const FabD out = outt[fab], v = vt[fab];
    const int vhx = min(tx0 + TX, b.hi[0] + 1), vhy = min(ty0 + TY, b.hi[1] + 1);      // last staged column / row (ghost included)
    typedef const __attribute__((address_space(1))) char gbyte;
    auto gat = [](const FabD::gdouble* plane, unsigned byteoff) -> FabD::gdouble& {
        return *(FabD::gdouble*)((gbyte*)plane + (size_t)byteoff);
    };
    // staging elements of this thread: 32-bit byte offsets inside a plane of v (a uniform plane pointer is added per plane)
    unsigned pfo[NE];
    bool pfl[NE];
#pragma unroll
    for (int s_ = 0; s_ < NE; ++s_) {
        const int e = tid + s_ * NT;
        const int ii = tx0 - 1 + e % W, jj = ty0 - 1 + e / W;
        pfl[s_] = e < PS && ii <= vhx && jj <= vhy;
        pfo[s_] = pfl[s_] ? 8u * (unsigned)((ii - v.lo[0]) + v.n[0] * (jj - v.lo[1])) : 0u;
    }
    const long vks = (long)v.n[0] * v.n[1];
    double pf[NE][3];
    auto fetch = [&](int k) {
        const FabD::gdouble* pl = v.gp() + vks * (k - v.lo[2]);
#pragma unroll
        for (int s_ = 0; s_ < NE; ++s_)
            if (pfl[s_]) {
#pragma unroll
                for (int n = 0; n < 3; ++n) pf[s_][n] = gat(pl + v.cs * n, pfo[s_]);
            }
    };
    auto commit = [&](int k) {
        double* dst = V[(k - k0 + 1) & 3];
#pragma unroll
        for (int s_ = 0; s_ < NE; ++s_)
            if (pfl[s_]) {
                const int e = tid + s_ * NT;
#pragma unroll
                for (int n = 0; n < 3; ++n) dst[e + PS * n] = pf[s_][n];
            }
    };
    fetch(k0 - 1); commit(k0 - 1);
    fetch(k0); commit(k0);
    fetch(k0 + 1); commit(k0 + 1);
    if (k0 + 2 <= k1 + 1) fetch(k0 + 2);
    // output / right-hand side / a-term: byte offsets of the cell inside a plane
    const unsigned oo = on ? 8u * (unsigned)((i - out.lo[0]) + out.n[0] * (j - out.lo[1])) : 0u;
    const long oks = (long)out.n[0] * out.n[1];
    FabD rh = out, ac = out;
    unsigned ro = 0, ao = 0;
    long rks = 0, aks = 0;
    if constexpr (RES) { rh = rhst[fab]; ro = on ? 8u * (unsigned)((i - rh.lo[0]) + rh.n[0] * (j - rh.lo[1])) : 0u; rks = (long)rh.n[0] * rh.n[1]; }
    if constexpr (HASA) { ac = at[fab]; ao = on ? 8u * (unsigned)((i - ac.lo[0]) + ac.n[0] * (j - ac.lo[1])) : 0u; aks = (long)ac.n[0] * ac.n[1]; }
    const int tb = (li + 1) + W * (lj + 1);
    for (int k = k0; k <= k1; ++k) {
        double rv[3] = {0., 0., 0.}, av = 0.0;
        if (on) {
            if constexpr (RES) {
                const FabD::gdouble* pr = rh.gp() + rks * (k - rh.lo[2]);
#pragma unroll
                for (int n = 0; n < 3; ++n) rv[n] = gat(pr + rh.cs * n, ro);
            }
            if constexpr (HASA) av = gat(ac.gp() + aks * (k - ac.lo[2]), ao);
        }
        __syncthreads();
        const int s0 = (k - k0 + 1) & 3;
        const double* P0 = V[s0] + tb;
        const double* Pm = V[(s0 + 3) & 3] + tb;
        const double* Pp = V[(s0 + 1) & 3] + tb;
        if (on) {
            // mixed differences: u -> (C_xy, C_xz), v -> (C_xy, C_yz), w -> (C_xz, C_yz)
            const double cxy_u = (P0[W + 1] - P0[-W + 1]) - (P0[W - 1] - P0[-W - 1]);
            const double cxz_u = (Pp[1] - Pm[1]) - (Pp[-1] - Pm[-1]);
            const double cxy_v = (P0[PS + W + 1] - P0[PS - W + 1]) - (P0[PS + W - 1] - P0[PS - W - 1]);
            const double cyz_v = (Pp[PS + W] - Pm[PS + W]) - (Pp[PS - W] - Pm[PS - W]);
            const double cxz_w = (Pp[2 * PS + 1] - Pm[2 * PS + 1]) - (Pp[2 * PS - 1] - Pm[2 * PS - 1]);
            const double cyz_w = (Pp[2 * PS + W] - Pm[2 * PS + W]) - (Pp[2 * PS - W] - Pm[2 * PS - W]);
            const double cr[3] = { p.cx[0][0] * cxy_v + p.cx[0][1] * cxz_w, p.cx[1][0] * cxy_u + p.cx[1][1] * cyz_w, p.cx[2][0] * cxz_u + p.cx[2][1] * cyz_v };
            FabD::gdouble* po = out.gp() + oks * (k - out.lo[2]);
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const double c0 = P0[PS * n];
                const double ax = HASA ? p.alpha * av * c0 : 0.0;
                const double y = ax
                    - p.dhx * (p.b[n][0] * (P0[PS * n + 1] - c0) - p.b[n][0] * (c0 - P0[PS * n - 1]))
                    - p.dhy * (p.b[n][1] * (P0[PS * n + W] - c0) - p.b[n][1] * (c0 - P0[PS * n - W]))
                    - p.dhz * (p.b[n][2] * (Pp[PS * n] - c0) - p.b[n][2] * (c0 - Pm[PS * n]));
                const double base = RES ? rv[n] - y : y;
                const double o = base + cr[n];
                gat(po + out.cs * n, oo) = o;
                const double ab = fabs(o);
                mx = fmax(mx, ab == ab ? ab : INFINITY);
            }
        }
        // plane k + 2 into the slot of plane k - 2 (every wavefront is past this plane's barrier, i.e. done with plane k - 1's arithmetic)
        if (k + 2 <= k1 + 1) {
            commit(k + 2);
            if (k + 3 <= k1 + 1) fetch(k + 3);
        }
    }
    if (normout) tnorm_commit(mx, normout);
#endif
}

template <int TX, int TY>
static void tensor_uni_launch(const Layout& l, const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& vel, const MultiFab* rhs,
                              unsigned long long* d_norm)
{
    const int ntx = (l.max_len[0] + TX - 1) / TX, nty = (l.max_len[1] + TY - 1) / TY;
    const int kcs_t = (int)tune("TENSOR_UNI_KC", 0);
    const int kcs = kcs_t > 0 ? kcs_t : std::min(32, std::max(4, l.max_len[2] / 8));
    const int nkc = (l.max_len[2] + kcs - 1) / kcs;
    const int total = ntx * nty * nkc;
    const int xcd_cnt = total >= 64 ? (total + 7) / 8 : 0;
    dim3 grid((unsigned)(xcd_cnt > 0 ? 8 * xcd_cnt : total), (unsigned)l.nlocal());
    TensorUniArgs p;
    const bool hasa = c.a && c.alpha != 0.0;
    const double sbeta = (rhs ? -1.0 : 1.0) * c.beta;
    const double hi[3] = {1.0 / g.dx[0], 1.0 / g.dx[1], 1.0 / g.dx[2]};
    p.alpha = c.alpha;
    p.dhx = c.beta / (g.dx[0] * g.dx[0]); p.dhy = c.beta / (g.dx[1] * g.dx[1]); p.dhz = c.beta / (g.dx[2] * g.dx[2]);
    for (int n = 0; n < 3; ++n)
        for (int d = 0; d < 3; ++d) p.b[n][d] = c.bu[d] * (n == d ? 4.0 / 3.0 : 1.0);
    // component n, its two partner directions (a < b, both != n): mixed difference in the plane (n, a) of component a, in (n, b) of component b
    for (int n = 0; n < 3; ++n) {
        const int a = n == 0 ? 1 : 0, bq = n == 2 ? 1 : 2;
        p.cx[n][0] = sbeta * (0.25 * hi[n] * hi[a]) * ((2.0 / 3.0) * c.bu[n] - c.bu[a]);
        p.cx[n][1] = sbeta * (0.25 * hi[n] * hi[bq]) * ((2.0 / 3.0) * c.bu[n] - c.bu[bq]);
    }
    auto& ctx = Context::get();
#define IAMRX_TU(R, A) hipLaunchKernelGGL((k_tensor_uni<TX, TY, R, A>), grid, dim3(TX * TY), 0, ctx.stream, l.d_boxes, out.d_tab, vel.d_tab, \
                                          rhs ? rhs->d_tab : nullptr, hasa ? c.a->d_tab : nullptr, p, d_norm, ntx, nty, nkc, kcs, xcd_cnt)
    if (rhs) { if (hasa) IAMRX_TU(true, true); else IAMRX_TU(true, false); }
    else { if (hasa) IAMRX_TU(false, true); else IAMRX_TU(false, false); }
#undef IAMRX_TU
}

// out = (rhs - A vel | A vel) for the constant-viscosity tensor operator in one launch (k_tensor_cross_zm<.., FUSE>); false: not applicable,
// the caller runs abec_residual (7-point part, then tensor_cross_terms_sub).  norm_out as in abec_residual.
bool tensor_residual_fused(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& vel, const MultiFab* rhs, double* norm_out)
{
    const Layout& l = *out.layout;
    if (tune("TENSOR_FUSED", 1) == 0 || tune("TENSOR_CROSS_ZM", 1) == 0 || tune("ABEC_SIG", 1) == 0) return false;
    if (!c.tensor || !c.tensor_eta || !c.b_uniform || vel.ncomp != 3 || vel.ngrow < 1 || l.max_len[0] < 32 || l.max_len[1] < 8) return false;
    auto& ctx = Context::get();
    static unsigned long long* d_norm = nullptr;
    if (norm_out && !d_norm) IAMRX_HIP_CHECK(hipMalloc(&d_norm, 2 * sizeof(unsigned long long)));
    if (out.nlocal() > 0) {
        if (norm_out) IAMRX_HIP_CHECK(hipMemsetAsync(d_norm, 0, sizeof(unsigned long long), ctx.stream));
        // the cell-centred form (k_tensor_uni) unless IAMRX_TENSOR_UNI_CC = 0 asks for the face-flux form
        const int cc = (int)tune("TENSOR_UNI_CC", 1);
        if (cc != 0) {
            unsigned long long* dn = norm_out ? d_norm : nullptr;
            switch (cc) {
            case 2: tensor_uni_launch<32, 16>(l, g, c, out, vel, rhs, dn); break;
            case 3: tensor_uni_launch<64, 8>(l, g, c, out, vel, rhs, dn); break;
            case 4: tensor_uni_launch<64, 4>(l, g, c, out, vel, rhs, dn); break;
            case 5: tensor_uni_launch<16, 16>(l, g, c, out, vel, rhs, dn); break;
            default: tensor_uni_launch<32, 8>(l, g, c, out, vel, rhs, dn); break;
            }
        } else {
        constexpr int TX = 32, TY = 8;
        const int ntx = (l.max_len[0] + TX - 1) / TX, nty = (l.max_len[1] + TY - 1) / TY;
        const int kcs = std::min(32, std::max(4, l.max_len[2] / 8));
        const int nkc = (l.max_len[2] + kcs - 1) / kcs;
        const int total = ntx * nty * nkc;
        const int xcd_cnt = total >= 64 ? (total + 7) / 8 : 0;
        dim3 grid((unsigned)(xcd_cnt > 0 ? 8 * xcd_cnt : total), (unsigned)l.nlocal());
        EtaUni eu;
        for (int d = 0; d < 3; ++d) eu.v[d] = c.bu[d];
        FuseArgs fa;
        fa.rhst = rhs ? rhs->d_tab : nullptr;
        fa.at = (c.a && c.alpha != 0.0) ? c.a->d_tab : nullptr;
        fa.alpha = c.alpha;
        fa.dhx = c.beta / (g.dx[0] * g.dx[0]); fa.dhy = c.beta / (g.dx[1] * g.dx[1]); fa.dhz = c.beta / (g.dx[2] * g.dx[2]);
        hipLaunchKernelGGL((k_tensor_cross_zm<true, TX, TY, true, true>), grid, dim3(TX * TY), 0, ctx.stream, l.d_boxes, out.d_tab, vel.d_tab,
                           c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab, 1.0 / g.dx[0], 1.0 / g.dx[1], 1.0 / g.dx[2], (rhs ? -1.0 : 1.0) * c.beta,
                           norm_out ? d_norm : nullptr, ntx, nty, nkc, kcs, xcd_cnt, eu, fa);
        }
    }
    if (norm_out) {
        double v = 0.0;
        const bool global = !l.replicated && ctx.comm->nranks > 1;
        if (out.nlocal() == 0 && global) IAMRX_HIP_CHECK(hipMemsetAsync(d_norm, 0, sizeof(unsigned long long), ctx.stream));
        if (global) ctx.comm->allreduce_device(reinterpret_cast<double*>(d_norm), 1, ReduceOp::Max, ctx.stream);
        if (out.nlocal() > 0 || global) {
            unsigned long long bits = 0;
            IAMRX_HIP_CHECK(hipMemcpyAsync(&bits, d_norm, sizeof(bits), hipMemcpyDeviceToHost, ctx.stream));
            ctx.sync();
            std::memcpy(&v, &bits, sizeof(v));
        }
        *norm_out = v;
    }
    return true;
}

// out += sign * beta * div(cross fluxes)
void tensor_cross_terms_sub(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& vel, double sign, unsigned long long* normout)
{
    if (out.nlocal() == 0) return;
    IAMRX_ASSERT(vel.ncomp == 3 && c.b[0]->ncomp == (c.tensor_eta ? 1 : 3));
    const Layout& l = *out.layout;
    const bool zm = tune("TENSOR_CROSS_ZM", 1) != 0;
    if (zm && l.max_len[0] >= 32 && l.max_len[1] >= 8 && vel.ngrow >= 1) {
        constexpr int TX = 32, TY = 8;
        const int ntx = (l.max_len[0] + TX - 1) / TX, nty = (l.max_len[1] + TY - 1) / TY;
        const int kcs = std::min(32, std::max(4, l.max_len[2] / 8));
        const int nkc = (l.max_len[2] + kcs - 1) / kcs;
        const int total = ntx * nty * nkc;
        const int xcd_cnt = total >= 64 ? (total + 7) / 8 : 0;
        dim3 grid((unsigned)(xcd_cnt > 0 ? 8 * xcd_cnt : total), (unsigned)l.nlocal());
#define IAMRX_TCZ(E) hipLaunchKernelGGL((k_tensor_cross_zm<E, TX, TY>), grid, dim3(TX * TY), 0, Context::get().stream, l.d_boxes, out.d_tab, vel.d_tab, \
                       c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab, 1.0 / g.dx[0], 1.0 / g.dx[1], 1.0 / g.dx[2], sign * c.beta, normout, ntx, nty, nkc, kcs, xcd_cnt)
        if (c.tensor_eta && c.b_uniform && tune("ABEC_SIG", 1) != 0) {
            EtaUni eu;
            for (int d = 0; d < 3; ++d) eu.v[d] = c.bu[d];
            hipLaunchKernelGGL((k_tensor_cross_zm<true, TX, TY, true>), grid, dim3(TX * TY), 0, Context::get().stream, l.d_boxes, out.d_tab, vel.d_tab,
                               c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab, 1.0 / g.dx[0], 1.0 / g.dx[1], 1.0 / g.dx[2], sign * c.beta, normout, ntx, nty, nkc, kcs,
                               xcd_cnt, eu);
        } else if (c.tensor_eta) IAMRX_TCZ(true); else IAMRX_TCZ(false);
#undef IAMRX_TCZ
        return;
    }
    Tiling t = level_tiling(*out.layout, cell_type(), 0, 8);
    if (c.tensor_eta) {
        hipLaunchKernelGGL(k_tensor_cross<true>, t.grid(), Tiling::block(), 0, Context::get().stream, t, out.layout->d_boxes, out.d_tab, vel.d_tab,
                           c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab, 1.0 / g.dx[0], 1.0 / g.dx[1], 1.0 / g.dx[2], sign * c.beta, normout);
        return;
    }
    hipLaunchKernelGGL(k_tensor_cross<false>, t.grid(), Tiling::block(), 0, Context::get().stream, t, out.layout->d_boxes, out.d_tab, vel.d_tab,
                       c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab, 1.0 / g.dx[0], 1.0 / g.dx[1], 1.0 / g.dx[2], sign * c.beta, normout);
}

// MLTensorOp::setShearViscosity: b_d(comp) = eta_d * (comp == d ? 4/3 : 1), bulk viscosity 0
// Diffusion::computeExtensiveFluxes on the tensor operator (Source/Diffusion.cpp:1463-1537: MLMG::getFluxes, i.e. the face fluxes of
// the operator without its b scalar, times fac x face area): flux_d(n) = fac * area_d * ( -eta_d (4/3 if n == d) du_n/dx_d + cross_d(n) ).
// vel: 3 comps, the ghost cells hold what the operator put there (CellMG::applyBC); eta: 1-component face viscosity; flux_d: 3 comps
template <int D>
static void tensor_flux_dir(const Geometry& g, const MultiFab& vel, const MultiFab& eta, MultiFab& flux, double fac, bool add)
{
    const FabD *vt = vel.d_tab, *et = eta.d_tab, *ft = flux.d_tab;
    const double dxi = 1.0 / g.dx[0], dyi = 1.0 / g.dx[1], dzi = 1.0 / g.dx[2];
    const double hinv = 1.0 / g.dx[D];
    const double scale = fac * g.dx[(D + 1) % 3] * g.dx[(D + 2) % 3];
    for_each(*vel.layout, face_type(D), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD v = vt[f], e = et[f];
        double cf[3];
        cross_flux<D, true>(v, e, i, j, k, dxi, dyi, dzi, cf);
        const int im = i - (D == 0), jm = j - (D == 1), km = k - (D == 2);
        const double e1 = e(i, j, k, 0);
        for (int n = 0; n < 3; ++n) {
            const double b = e1 * (n == D ? 4.0 / 3.0 : 1.0);
            const double fl = scale * (-b * (v(i, j, k, n) - v(im, jm, km, n)) * hinv + cf[n]);
            if (add) ft[f](i, j, k, n) += fl; else ft[f](i, j, k, n) = fl;
        }
    });
}
void tensor_extensive_flux(const Geometry& g, const MultiFab& vel, const MultiFab* const eta[3], MultiFab* const flux[3], double fac, bool add)
{
    if (vel.nlocal() == 0) return;
    IAMRX_ASSERT(vel.ncomp >= 3 && vel.ngrow >= 1 && eta[0]->ncomp == 1);
    tensor_flux_dir<0>(g, vel, *eta[0], *flux[0], fac, add);
    tensor_flux_dir<1>(g, vel, *eta[1], *flux[1], fac, add);
    tensor_flux_dir<2>(g, vel, *eta[2], *flux[2], fac, add);
}

void tensor_bcoef(MultiFab& b3, const MultiFab& eta, int dir)
{
    if (b3.nlocal() == 0) return;
    const FabD *bt = b3.d_tab, *et = eta.d_tab;
    for_each(*b3.layout, face_type(dir), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const double e = et[f](i, j, k, 0);
        for (int n = 0; n < 3; ++n) bt[f](i, j, k, n) = e * (n == dir ? 4.0 / 3.0 : 1.0);
    });
}

// Edge/corner ghost cells of the velocity needed by the cross terms (MLTensorOp::applyBCTensor ->
// mltensor_fill_edges / mltensor_fill_corners).  Fully periodic or interior edges are filled by FillBoundary.  A cell
// outside the box in >= 2 directions of which at least one is a non-periodic domain face gets the average over those
// exterior directions of the one-dimensional face rule applied along that direction (Neumann: neighbour copy;
// Dirichlet: the face extrapolation polynomial on the already filled ghost cells + boundary value of bcval).
struct EdgeDesc { int fab; BoxD region; int sgn[3]; int ext[3]; };
struct EdgeParams { int bclo[3], bchi[3]; int NX[3]; double c[3][4]; int inhomog; int ncomp; int comp0; };

__global__ void __launch_bounds__(256) k_tensor_edges(const EdgeDesc* __restrict__ descs, const FabD* __restrict__ pt,
                                                      const FabD* __restrict__ bvt, EdgeParams P)
{
    const EdgeDesc ed = descs[blockIdx.y];
    const FabD phi = pt[ed.fab];
    const int nx = ed.region.len(0), ny = ed.region.len(1);
    const long npts = ed.region.npts();
    const int next = ed.ext[0] + ed.ext[1] + ed.ext[2];
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        int idx[3];
        idx[0] = ed.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        idx[1] = ed.region.lo[1] + (int)(r % ny);
        idx[2] = ed.region.lo[2] + (int)(r / ny);
        for (int n = P.comp0; n < P.comp0 + P.ncomp; ++n) {
            double sum = 0.0;
            for (int d = 0; d < 3; ++d) {
                if (!ed.ext[d]) continue;
                const int bct = ed.sgn[d] < 0 ? P.bclo[d] : P.bchi[d];
                const int s = ed.sgn[d] < 0 ? 1 : -1;
                double v;
                int m3[3] = {idx[0], idx[1], idx[2]};
                if (bct == lo_neumann) { m3[d] += s; v = phi(m3[0], m3[1], m3[2], n); }
                else if (bct == lo_reflect_odd) { m3[d] += s; v = -phi(m3[0], m3[1], m3[2], n); }
                else {
                    const double bv = (P.inhomog && bvt) ? bvt[ed.fab](idx[0], idx[1], idx[2], n) : 0.0;
                    if (P.NX[d] < 2) v = bv;
                    else {
                        double tmp = 0.0;
                        for (int m = 1; m < P.NX[d]; ++m) { m3[d] = idx[d] + m * s; tmp += phi(m3[0], m3[1], m3[2], n) * P.c[d][m]; }
                        v = tmp + bv * P.c[d][0];
                    }
                }
                sum += v;
            }
            phi(idx[0], idx[1], idx[2], n) = sum / (double)next;
        }
    }
}

void fill_tensor_corners(const Geometry& g, MultiFab& phi, const DomainBC& bc, bool inhomog, const MultiFab* bcval, int comp0, int ncomp)
{
    if (ncomp < 0) ncomp = phi.ncomp - comp0;
    bool anywall = false;
    for (int d = 0; d < 3; ++d) if (!g.periodic[d]) anywall = true;
    if (!anywall || phi.nlocal() == 0) return;
    auto& ctx = Context::get();
    EdgeParams P;
    for (int d = 0; d < 3; ++d) {
        P.bclo[d] = bc.lo[d]; P.bchi[d] = bc.hi[d];
        const int blen = g.domain.len(d);
        P.NX[d] = blen + 1 < bc.maxorder ? blen + 1 : bc.maxorder;
        const double x[4] = {0.0, 0.5, 1.5, 2.5};
        for (int j = 0; j < 4; ++j) P.c[d][j] = 0.0;
        for (int j = 0; j < P.NX[d]; ++j) {
            double num = 1.0, den = 1.0;
            for (int i = 0; i < P.NX[d]; ++i) { if (i == j) continue; num *= -0.5 - x[i]; den *= x[j] - x[i]; }
            P.c[d][j] = num / den;
        }
    }
    P.inhomog = inhomog ? 1 : 0;
    P.ncomp = ncomp; P.comp0 = comp0;
    for (int nout = 2; nout <= 3; ++nout) {
        static auto& cache = make_desc_cache<EdgeDesc>();
        int nd; long maxpts;
        const EdgeDesc* dd = cached_descs(cache, {(long)phi.layout->id, nout, g.domain.lo[0], g.domain.lo[1], g.domain.lo[2], g.domain.hi[0],
                                                  g.domain.hi[1], g.domain.hi[2], g.periodic[0] + 2 * g.periodic[1] + 4 * g.periodic[2], 0},
            [&](std::vector<EdgeDesc>& descs, long& mp) {
        for (int li = 0; li < phi.nlocal(); ++li) {
            const BoxD vb = phi.layout->lbox(li);
            for (int sz = -1; sz <= 1; ++sz) for (int sy = -1; sy <= 1; ++sy) for (int sx = -1; sx <= 1; ++sx) {
                const int sg[3] = {sx, sy, sz};
                if ((sx != 0) + (sy != 0) + (sz != 0) != nout) continue;
                EdgeDesc e;
                e.fab = li;
                int next = 0;
                for (int d = 0; d < 3; ++d) {
                    e.sgn[d] = sg[d];
                    e.region.lo[d] = sg[d] < 0 ? vb.lo[d] - 1 : (sg[d] > 0 ? vb.hi[d] + 1 : vb.lo[d]);
                    e.region.hi[d] = sg[d] < 0 ? vb.lo[d] - 1 : (sg[d] > 0 ? vb.hi[d] + 1 : vb.hi[d]);
                    const bool at_face = sg[d] < 0 ? vb.lo[d] == g.domain.lo[d] : (sg[d] > 0 ? vb.hi[d] == g.domain.hi[d] : false);
                    e.ext[d] = (sg[d] != 0 && !g.periodic[d] && at_face) ? 1 : 0;
                    next += e.ext[d];
                }
                if (next == 0) continue;
                descs.push_back(e);
                mp = std::max(mp, e.region.npts());
            }
        }
            }, nd, maxpts);
        if (nd == 0) continue;
        long nb = (maxpts + 255) / 256; if (nb > 16) nb = 16; if (nb < 1) nb = 1;
        hipLaunchKernelGGL(k_tensor_edges, dim3((unsigned)nb, (unsigned)nd), dim3(256), 0, ctx.stream, dd, phi.d_tab,
                           bcval ? bcval->d_tab : nullptr, P);
    }
}

}  // namespace iamrx
