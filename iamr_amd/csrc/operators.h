// iamr_amd/csrc/operators.h -- host-side operator layer keeping IAMR's operator API surface
// (MacProj / Projection / Diffusion / NavierStokesBase), SURVEY 8(b).
#pragma once
#include "mf.h"
#include "mlmg.h"
#include "kernels.h"

namespace iamrx {

// MacProj::mlmg_mac_solve (reference Source/MacProj.H:82-94, Source/MacProj.cpp:1084-1184)
MGStats mlmg_mac_solve(const Geometry& g, MultiFab* const umac[3], const MultiFab& rho, int rho_comp, const MultiFab* S,
                       MultiFab& mac_phi, double rhs_scale, const DomainBC& bc, double mac_tol, double mac_abs_tol,
                       const MGOpts& opts, MultiFab* const fluxes[3]);

}  // namespace iamrx
