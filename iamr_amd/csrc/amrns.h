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
                              const MultiFab* rhnd, double rtol, double atol, bool increment_gp, double inflow_scale, const MultiFab* const rhcc[] = nullptr);
    ProjLevel proj_level(int l);                 // the descriptor of level l for composite_project / ml_sync_project
    // building blocks, public for the unit tests
    void reflux(int l);
    void avg_down(int l);
    void mac_sync(int l);
    void level_sync(int l, int crse_iteration = -1);      // crse_iteration: of level l within the step of level l-1 (-1: the last one)
    void post_timestep(int l, int crse_iteration = -1);
    void time_step(int l, double time, int iteration, int niter);
    // ---- regridding (Amr::regrid from level 0 at the start of a coarse step; NavierStokes::errorEst, NS_error.cpp:10-145;
    // NavierStokesBase::init(AmrLevel&) / init(), NavierStokesBase.cpp:1713-1806) ----
    struct TagRule {
        int comp = Tracer;              // state component, or -1: magnitude of vorticity (mag_vort)
        int mode = 0;                   // 0 value_greater, 1 value_less, 2 vorticity_greater (x 2^level), 3 adjacent_difference_greater
        std::vector<double> value;      // per level (the last one repeats)
        int max_level = 1000;           // tags only on levels < max_level
        bool has_box = false;
        double box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};
    };
    struct RegridOpts {
        int max_level = 0, regrid_int = 0;
        int blocking_factor = 8, max_grid_size = 32, n_error_buf = 1;
        double grid_eff = 0.7;
        int do_refine_outflow = 0, do_derefine_outflow = 1, nbuf_outflow = 1;   // ns.do_refine_outflow / do_derefine_outflow / Nbuf_outflow (NavierStokesBase.cpp:136-138)
        int compute_new_dt_on_regrid = 0;   // amr.compute_new_dt_on_regrid (Amr::timeStep: computeNewDt(post_regrid_flag = 1) after a level-0 regrid; default 0)
        std::vector<TagRule> rules;
    };
    void set_regrid(const RegridOpts& r) { const int keep = rg.compute_new_dt_on_regrid; rg = r; if (!r.compute_new_dt_on_regrid) rg.compute_new_dt_on_regrid = keep; }
    void set_compute_new_dt_on_regrid(int on) { rg.compute_new_dt_on_regrid = on; }
    void set_outflow_tagging(int refine, int derefine, int nbuf) { rg.do_refine_outflow = refine; rg.do_derefine_outflow = derefine; rg.nbuf_outflow = nbuf; }
    // new grids of levels 1 .. max_level from the tags of the current data (coarse to fine nesting enforced); the level-l boxes
    std::vector<std::vector<BoxD>> make_new_grids(int lbase = 0);
    // install grids (levels 1 ..): new levels are filled from the old level where it existed and from the next coarser level elsewhere;
    // returns false if nothing changed
    bool install_grids(const std::vector<std::vector<BoxD>>& grids, int lbase = 0, double cur_time = 0.0);
    bool regrid(int lbase = 0, double cur_time = 0.0) { return install_grids(make_new_grids(lbase), lbase, cur_time); }
    // Amr::timeStep's regrid check (okToRegrid(i) for i = level .. finest at the start of every step of every level): level_count_v[i] = steps
    // of level i since the grids above it were last rebuilt.  regrid_log: what was rebuilt during the last coarse step (base level, time,
    // boxes per level) -- a driver that mirrors the hierarchy elsewhere (the tests' oracle) replays it
    std::vector<int> level_count_v;
    struct RegridEvent { int lbase; double time; std::vector<std::vector<BoxD>> grids; };
    std::vector<RegridEvent> regrid_log;
    void maybe_regrid(int l, double time);
    int level_count = 0;                // coarse steps since the last regrid (Amr::level_count[0])
    // Amr::checkPoint / restart: dt_level, dt_min, n_cycle per level, level_steps, level_count, stop_time
    void get_restart_state(double* dt_lev, double* dt_mn, int* ncyc, int counters[2], double* stop) const;
    void set_restart_state(const double* dt_lev, const double* dt_mn, const int* ncyc, const int counters[2], double stop);
    void get_level_counts(int* counts, int n) const;
    void set_level_counts(const int* counts, int n);
    uint64_t grid_generation() const { return m_grid_gen; }   // incremented whenever the grids change
    uint64_t m_grid_gen = 0;
    // section profile of the coarse step (host clock around stream syncs, only while profile_on): [0] reflux, [1] avgDown,
    // [2] mac_sync_solve, [3] mac_sync rest (re-advection, viscous / scalar sync solves, SyncInterp), [4] level_sync (MLsyncProject +
    // interpolation to finer levels), [5] regrid, [8 + l] advance of level l (all its sub-steps; the level's own t_sections split it further)
    bool profile_on = false;
    double t_prof[16] = {0};
    void set_profile(bool on);

private:
    NSParams p;
    MGOpts o;
    std::vector<std::unique_ptr<NavierStokes>> lev;
    std::vector<int> n_cycle;
    std::vector<double> dt_level, dt_min;
    int level_steps = 0;
    double stop_time = -1.0;
    RegridOpts rg;
    int m_ratio = 2;
    void link_level(int l);
    void check_nesting(const std::vector<BoxD>& fine, const std::vector<BoxD>& crse, const Geometry& cgeom, int l) const;
    void validate_grids(const std::vector<std::vector<BoxD>>& grids, int lbase = 0) const;   // throws before anything is modified
    void compute_new_dt(bool post_regrid);
};

}  // namespace iamrx
