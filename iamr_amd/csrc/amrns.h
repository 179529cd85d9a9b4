// iamr_amd/csrc/amrns.h -- the AMR hierarchy driver (amrns.hip): the role of amrex::Amr::coarseTimeStep / timeStep for IAMR's
// NavierStokes levels, NavierStokesBase::post_timestep and the multi-level projections of Projection.
#pragma once
#include "operators.h"
#include <vector>
#include <memory>

namespace iamrx {

class AmrNS {
public:
    // layouts[l]: boxes of level l in that level's index space (level 0 covers the domain); ratio 2 between levels
    AmrNS(const Geometry& g0, const std::vector<LayoutP>& layouts, int ratio, const NSParams& p, const MGOpts& o);
    int nlevels() const { return (int)lev.size(); }
    NavierStokes& level(int l) { return *lev[l]; }
    void post_init(double stop_time);            // NavierStokes::post_init for the hierarchy (Source/NavierStokes.cpp:1254-1299)
    double coarse_step();                        // Amr::coarseTimeStep; returns the level-0 dt
    double time() const { return lev[0]->time; }
    double dt(int l) const { return dt_level[l]; }
    MGStats st_sync, st_mac_sync;                // last MLsyncProject / mac_sync_solve
    // Hydro::NodalProjector::project on levels c0 .. c0+nl-1 (Projection::doMLMGNodalProjection with nlevel > 1)
    MGStats composite_project(int c0, int nl, MultiFab* const vel[], const int vcomp[], MultiFab* const phi[], const MultiFab* const sig[],
                              const MultiFab* rhnd, double rtol, double atol, bool increment_gp, double inflow_scale);
    // building blocks, public for the unit tests
    void reflux(int l);
    void avg_down(int l);
    void mac_sync(int l);
    void level_sync(int l, int crse_iteration = -1);      // crse_iteration: of level l within the step of level l-1 (-1: the last one)
    void post_timestep(int l, int crse_iteration = -1);
    void time_step(int l, double time, int iteration, int niter);

private:
    NSParams p;
    MGOpts o;
    std::vector<std::unique_ptr<NavierStokes>> lev;
    std::vector<int> n_cycle;
    std::vector<double> dt_level, dt_min;
    int level_steps = 0;
    double stop_time = -1.0;
};

}  // namespace iamrx
