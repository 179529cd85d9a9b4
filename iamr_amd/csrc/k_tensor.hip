// iamr_amd/csrc/k_tensor.hip -- cross-derivative terms of the viscous stress tensor
// div( eta (grad u + grad u^T) - 2/3 eta (div u) I ) on top of the 3-component ABec operator
// (b_d(comp) = eta * (comp == d ? 4/3 : 1)).  Role: AMReX MLTensorOp (mltensor_cross_terms_f{x,y,z},
// mltensor_cross_terms) as used at reference Source/Diffusion.cpp:715-768 (explicit apply) and
// :858-923 (implicit solve); SURVEY a10, a11.
#include "kernels.h"
#include "launch.h"

namespace iamrx {

// cross flux on the d-face (i,j,k) for component n.  eta_n = b_d(comp d) * 3/4 (normal), eta_t = b_d(comp != d)
template <int D>
__device__ __forceinline__ void cross_flux(const FabD& v, const FabD& eta, int i, int j, int k, double dxi, double dyi, double dzi, double f[3])
{
    constexpr double twoThirds = 2.0 / 3.0;
    if (D == 0) {
        const double dudy = (v(i, j + 1, k, 0) + v(i - 1, j + 1, k, 0) - v(i, j - 1, k, 0) - v(i - 1, j - 1, k, 0)) * (0.25 * dyi);
        const double dvdy = (v(i, j + 1, k, 1) + v(i - 1, j + 1, k, 1) - v(i, j - 1, k, 1) - v(i - 1, j - 1, k, 1)) * (0.25 * dyi);
        const double dudz = (v(i, j, k + 1, 0) + v(i - 1, j, k + 1, 0) - v(i, j, k - 1, 0) - v(i - 1, j, k - 1, 0)) * (0.25 * dzi);
        const double dwdz = (v(i, j, k + 1, 2) + v(i - 1, j, k + 1, 2) - v(i, j, k - 1, 2) - v(i - 1, j, k - 1, 2)) * (0.25 * dzi);
        const double divu = dvdy + dwdz;
        const double xif = 0.0;
        const double mun = 0.75 * (eta(i, j, k, 0) - xif), mut = eta(i, j, k, 1);
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
        const double mun = 0.75 * (eta(i, j, k, 1) - xif), mut = eta(i, j, k, 0);
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
        const double mun = 0.75 * (eta(i, j, k, 2) - xif), mut = eta(i, j, k, 0);
        f[0] = -mut * dwdx;
        f[1] = -mut * dwdy;
        f[2] = -mun * (-twoThirds * divu) - xif * divu;
    }
}

__global__ void __launch_bounds__(256) k_tensor_cross(Tiling t, const BoxD* __restrict__ boxes, const FabD* __restrict__ outt,
    const FabD* __restrict__ vt, const FabD* __restrict__ ext, const FabD* __restrict__ eyt, const FabD* __restrict__ ezt,
    double dxi, double dyi, double dzi, double sbeta)
{
    const int fab = blockIdx.y;
    int i, j, k0, k1;
    if (!tile_ijk(t, boxes[fab], i, j, k0, k1)) return;
    const FabD out = outt[fab], v = vt[fab], ex = ext[fab], ey = eyt[fab], ez = ezt[fab];
    double fzl[3];
    cross_flux<2>(v, ez, i, j, k0, dxi, dyi, dzi, fzl);
    for (int k = k0; k <= k1; ++k) {
        double fxl[3], fxh[3], fyl[3], fyh[3], fzh[3];
        cross_flux<0>(v, ex, i, j, k, dxi, dyi, dzi, fxl);
        cross_flux<0>(v, ex, i + 1, j, k, dxi, dyi, dzi, fxh);
        cross_flux<1>(v, ey, i, j, k, dxi, dyi, dzi, fyl);
        cross_flux<1>(v, ey, i, j + 1, k, dxi, dyi, dzi, fyh);
        cross_flux<2>(v, ez, i, j, k + 1, dxi, dyi, dzi, fzh);
        for (int n = 0; n < 3; ++n) {
            out(i, j, k, n) += sbeta * (dxi * (fxh[n] - fxl[n]) + dyi * (fyh[n] - fyl[n]) + dzi * (fzh[n] - fzl[n]));
            fzl[n] = fzh[n];
        }
    }
}

// out += sign * beta * div(cross fluxes)
void tensor_cross_terms_sub(const Geometry& g, const AbecCoef& c, MultiFab& out, const MultiFab& vel, double sign)
{
    if (out.nlocal() == 0) return;
    IAMRX_ASSERT(vel.ncomp == 3 && c.b[0]->ncomp == 3);
    Tiling t = level_tiling(*out.layout, cell_type(), 0, 8);
    hipLaunchKernelGGL(k_tensor_cross, t.grid(), Tiling::block(), 0, Context::get().stream, t, out.layout->d_boxes, out.d_tab, vel.d_tab,
                       c.b[0]->d_tab, c.b[1]->d_tab, c.b[2]->d_tab, 1.0 / g.dx[0], 1.0 / g.dx[1], 1.0 / g.dx[2], sign * c.beta);
}

// MLTensorOp::setShearViscosity: b_d(comp) = eta_d * (comp == d ? 4/3 : 1), bulk viscosity 0
void tensor_bcoef(MultiFab& b3, const MultiFab& eta, int dir)
{
    if (b3.nlocal() == 0) return;
    const FabD *bt = b3.d_tab, *et = eta.d_tab;
    for_each(*b3.layout, face_type(dir), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const double e = et[f](i, j, k, 0);
        for (int n = 0; n < 3; ++n) bt[f](i, j, k, n) = e * (n == dir ? 4.0 / 3.0 : 1.0);
    });
}

// edge/corner ghost cells of the velocity needed by the cross terms.  Fully periodic levels get them
// from FillBoundary; wall-bounded levels: TODO(next): mltensor_fill_edges/corners.
void fill_tensor_corners(const Geometry& g, MultiFab& phi, const DomainBC& bc)
{
    (void)phi; (void)bc;
    for (int d = 0; d < 3; ++d)
        if (!g.periodic[d]) throw Error("iamrx: tensor operator with non-periodic boundaries not implemented yet");
}

}  // namespace iamrx
