// iamr_amd/csrc/mlmg.h -- geometric multigrid drivers (host side) for the cell-centred ABec/tensor
// operator and the nodal Laplacian.  Plays the role of amrex::MLMG + MLABecLaplacian/MLTensorOp/
// MLNodeLaplacian as used at reference Source/MacProj.cpp:1150-1183, Source/Diffusion.cpp:715-923,
// Source/Projection.cpp:2512-2542.
#pragma once
#include <functional>
#include "mf.h"
#include "kernels.h"
#include <vector>
#include <memory>

namespace iamrx {

// The fp64 round-off floor of a residual (about 1e-12 of the right-hand side once h <= 1/256) can sit just above the requested tolerance.
// hist: the residual norms after each cycle so far (hist.back() = the current one), r0: the norm in front of the first cycle.  Two signs
// of the floor, both only for a residual within 10x of the target:
//   (a) a COLLAPSE of the convergence rate -- the last cycle gained less than a factor 2 although some earlier cycle of this solve gained
//       more than a factor 5: a solve that converges slowly does so from its first cycle on, one that hits the floor stops from one cycle
//       to the next (round 6: one cycle to notice instead of three; IAMRX_MG_STALL_FAST = 0 keeps (b) only);
//   (b) less than 10 % lost over three cycles (rounds 3-5).
// amrex::MLMG has no such exit (it iterates to max_iter and aborts): the solvers report converged = 2 and a warning.
inline bool mg_stalled_at_floor(const std::vector<double>& hist, double r0, double target, bool fast)
{
    const size_t n = hist.size();
    if (n == 0 || !(hist[n - 1] <= 10.0 * target)) return false;
    if (n >= 4 && hist[n - 1] > 0.9 * hist[n - 4]) return true;
    if (!fast || n < 2) return false;
    const double last = hist[n - 1] / hist[n - 2];
    double best = 1.0, prev = r0;
    for (size_t i = 0; i + 1 < n; ++i) { if (prev > 0.0) best = std::min(best, hist[i] / prev); prev = hist[i]; }
    return last > 0.5 && best < 0.2;
}


struct MGOpts {
    int nu1 = 2, nu2 = 2, nuf = 8, nub = 0;
    int max_iters = 200;
    int bottom_maxiter = 200;
    double bottom_reltol = 1.e-4;
    double omega = 1.15;          // GSRB over-relaxation
    int maxorder = 3;
    int max_coarsening_level = 30;
    int min_width = 2;
    // Nodal cycle shape.  amrex::MLNodeLaplacian / MLMG use 4 sweeps per smooth call and nu1 = nu2 = 2 (16 sweeps on the finest level
    // per V-cycle); the converged solution does not depend on it, the time to solution does: at 256^3 on MI355X 2 sweeps x (1 + 1)
    // reaches 1e-11 in 19 ms against 30 ms (tools/bench_nodalmg.py), so that is the default here.
    int nodal_sweeps = 2;         // Gauss-Seidel sweeps per nodal smooth call
    int nodal_smoother = 0;       // 0: 8-colour Gauss-Seidel, 2: weighted Jacobi (2/3)
    int verbose = 0;
    int bottom_smoother_only = 0;
    int fixed_iters = 0;
    int nodal_nu1 = 1, nodal_nu2 = 1;   // pre / post smooth calls of the nodal V-cycle (nu1 / nu2 above: the cell-centred one)
    // cell-centred hierarchy: 1 = stop coarsening at the first single-box level of at most 8^3 cells and solve it with the
    // single-workgroup device BiCGStab (k_abec_bottom, no host synchronisation); 0 = coarsen to min_width and drive BiCGStab from the
    // host (upstream's shape; same converged solution, iteration counts may differ by one)
    int device_bottom = 1;
    int slab = 0;                 // 2-D problem on a thin periodic slab: keep the slab at two cells, coarsen the plane (mg_slab_level)
};

// V-cycle wall time without draining the stream: HIP events recorded in front of and behind every cycle on the launch stream; the
// durations are read after the solve (its last residual norm has synchronised the stream by then)
struct CycleTimer {
    std::vector<hipEvent_t> ev;
    int used = 0;
    void mark(hipStream_t s)
    {
        if (used == (int)ev.size()) { hipEvent_t e; IAMRX_HIP_CHECK(hipEventCreate(&e)); ev.push_back(e); }
        IAMRX_HIP_CHECK(hipEventRecord(ev[used++], s));
    }
    double total_ms()        // pairs (0,1), (2,3), ...; the events must have completed
    {
        double t = 0.0;
        for (int i = 0; i + 1 < used; i += 2) { float ms = 0.f; IAMRX_HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); t += ms; }
        used = 0;
        return t;
    }
};
CycleTimer& cycle_timer();

// multi-rank runs: MG levels whose total size is at most this many cells are replicated on every rank (one all-gather per
// V-cycle instead of a halo exchange per smoothing pass and an all-reduce per Krylov dot product); 0 disables
long mg_agglomeration_cells();
bool mg_agglomerate_level(const Layout& coarse);      // mlmg.hip
bool mg_slab_level(const Geometry& g, const Layout& l, int min_width, bool slab_problem);      // mlmg.hip: y kept at two cells from here on
Geometry mg_slab_geom(const Geometry& fine);


struct MGStats {
    int iters = 0;
    double resnorm0 = 0, rhsnorm0 = 0, resnorm = 0;
    int bottom_iters_total = 0;
    int converged = 0;
    double vcycle_ms = 0;         // mean wall time of one V-cycle (host clock around stream sync)
    int nlevels = 0;
};

class CellMG {
public:
    CellMG(const Geometry& g, LayoutP layout, int ncomp, const DomainBC& bc, const MGOpts& o);
    void setScalars(double alpha, double beta) { m_alpha = alpha; m_beta = beta; }
    void setACoeffs(const MultiFab* a) { m_a0 = a; }
    void setBCoeffs(const MultiFab* const b[3]) { for (int d = 0; d < 3; ++d) m_b0[d] = b[d]; }
    // the b of setBCoeffs is mac_bcoef(b, sig, comp, scale): the finest level's smoother and residual recompute it from sig (AbecCoef::sig)
    void setBCoeffsFromCell(const MultiFab* sig, int comp, double scale) { m_sig = sig; m_sig_comp = comp; m_sig_scale = scale; }
    void setTensor(bool t) { m_tensor = t; }
    // tensor operator with the B coefficients given as the 1-component face viscosity: the finest level streams eta instead of the
    // three-component b = eta * (4/3 on the normal component) arrays (a third of the coefficient traffic of the smoother)
    void setTensorEta(bool t) { m_tensor_eta = t; }
    // one DomainBC per component (MLTensorOp::setDomainBC with per-component arrays)
    void setDomainBCs(const DomainBC* bcs, int n) { m_bcn.assign(bcs, bcs + n); }
    // MLLinOp::setCoarseFineBC: the level does not cover the domain; crse = the coarse level's solution (cell centred, valid data on
    // the coarse level's own layout), cgeom its geometry.  crse == nullptr: homogeneous coarse/fine data.  Call before prepare().
    void setCoarseFineBC(const MultiFab* crse, const Geometry& cgeom, int ratio) { m_cf = true; m_crse = crse; m_cgeom = cgeom; m_ratio = ratio; }
    void prepare();   // build the coarse hierarchy (coefficient averaging)
    MGStats solve(MultiFab& phi, const MultiFab& rhs, double rtol, double atol);
    // out = L(phi) with inhomogeneous BC taken from phi's ghost cells
    void apply(MultiFab& out, MultiFab& phi);
    void fluxes(MultiFab& phi, MultiFab* const flux[3], MultiFab* const add_to[3]);
    int nlevels() const { return (int)m_lev.size(); }
    AbecCoef coef(int l) const;
    const Geometry& geom(int l) const { return m_lev[l].g; }
    void applyBC(int l, MultiFab& phi, bool inhomog, const MultiFab* bcval, bool corners = true);
    // cf_ghosts_current: the coarse/fine ghost cells are already what a fill would write (kept so by the passes themselves, k_abec.hip cf_maintain)
    void smooth(int l, MultiFab& sol, const MultiFab& rhs, bool skip_fill, bool cf_ghosts_current = false, bool sol_is_zero = false);
    bool zero_first_pass_ok(int l, const MultiFab& sol) const;   // the first colour pass can take the place of sol.setVal(0) (abec_gsrb_zero_ok)
    // nsweeps red+black sweeps; uses the fused out-of-place kernel (ping-pong with a level buffer) where it applies
    // acc (finest level, the last smoothing call of a V-cycle): the solution of the running solve; where the sweep kernel runs its last sweep
    // stores acc + correction into acc (m_acc_done is set: the caller skips its `sol += cor`; sol is then one sweep behind and unused)
    void smooth_n(int l, MultiFab& sol, const MultiFab& rhs, int nsweeps, bool skip_first_fill, bool sol_is_zero = false, MultiFab* acc = nullptr);
    bool fused_smoother_ok(int l) const;
    bool nbr_sweep_ok(int l, const MultiFab& sol, const MultiFab& rhs) const;
    bool cf_sweep_ok(int l, const MultiFab& sol) const;
    void vcycle(MGStats& st);
    MultiFab& res(int l) { return m_lev[l].res; }
    MultiFab& cor(int l) { return m_lev[l].cor; }

private:
    struct Level {
        Geometry g;
        LayoutP layout;
        MultiFab a, b[3];          // owned (coarse levels)
        MultiFab cor, res, rescor;
        MultiFab buf;              // second buffer of the fused (out-of-place) GSRB sweeps
        int wk_flag = -1;          // the colour passes of this level apply the domain walls themselves (abec_gsrb_walls_inkernel_ok; -1: not asked yet)
        bool res_filled = false;   // multi-box sweep kernel: the ghost layer of `res` is current (filled once per V-cycle)
        bool slab = false;         // slab level (mlmg.hip: mg_slab_level): two cells in y kept; virt = the one-plane coarsening of the level
        LayoutP virt;              // above, through which the restrictions go (vres: their target, duplicated into res / tmp_d)
        MultiFab vres;
        // agglomeration (multi-rank): from this level down every rank holds the whole level; dist = the distributed coarsening
        // of the level above, through which restriction results are gathered and corrections are picked out
        bool agg = false;
        LayoutP dist;
        MultiFab tmp_d;
        MultiFab cfm;              // coarse/fine mask (levels that do not cover the domain), see cf_build_mask
        CfTab cftab;
    };
    int bicgstab(int l, MultiFab& sol, const MultiFab& rhs, double eps_rel, double eps_abs, int& niters);
    void bottom_solve(MGStats& st);
    bool tail_fused() const;
    void cf_bcval(MultiFab& bcval);
    void subtract_mean(int l, MultiFab& mf);
    Geometry m_g;
    int m_ncomp;
    std::vector<DomainBC> m_bcn;
    MGOpts m_o;
    double m_alpha = 0.0, m_beta = 1.0;
    const MultiFab* m_a0 = nullptr;
    const MultiFab* m_b0[3] = {nullptr, nullptr, nullptr};
    bool m_buni = false;                 // the finest level's b arrays are constants (prepare() checks)
    bool m_buni_coarse = false;          // ... and the coarser levels use the constants too (IAMRX_MG_COARSE_UNIFORM, 1)
    double m_bu[3] = {0.0, 0.0, 0.0};
    const MultiFab* m_sig = nullptr;
    int m_sig_comp = 0;
    double m_sig_scale = 1.0;
    bool m_tensor = false;
    bool m_tensor_eta = false;
    bool m_singular = false;
    bool m_bottom_dev = false;    // the coarsest level is solved by k_abec_bottom (one single-workgroup launch, no host synchronisation)
    // tensor operator with constant viscosity whose coarsest level is one box of <= 27 cells: that level is solved DIRECTLY by one
    // single-workgroup launch (k_dense_bottom: M = alpha diag(a) + beta B assembled from the cached operator matrix B, Gauss-Jordan
    // with partial pivoting) instead of the host-driven BiCGStab (mlmg.hip: bottom_direct_prepare)
    MultiFab* m_acc = nullptr;     // solve(): the solution array the V-cycle's last sweep may add its correction to (IAMRX_MG_ACC_LAST_SWEEP)
    bool m_acc_done = false;
    bool m_bottom_direct = false;
    const double* m_dB = nullptr;  // the cached matrix (device, column-major, m_dN x m_dN)
    std::shared_ptr<double> m_dBh;            // keeps m_dB alive when the cache evicts its entry
    int m_dN = 0;
    void bottom_direct_prepare();
    void bottom_direct_solve();
    double m_dd_rho = 0.0;        // estimated contraction of one red-black sweep of a diagonally dominant operator (prepare())
    int m_dd_sweeps = 0;          // > 0: diagonally dominant operator solved by sweeps of the finest level only (prepare())
    // the finest level is several boxes covering the domain and takes the one-launch red + black sweep (k_abec_gsrb_rb<.., NBR>): its
    // correction carries two ghost layers (one exchange per sweep), its residual one, and the level keeps copies of the density with two
    // filled ghost layers / of the a-term with one (prepare())
    bool m_nbr = false;
    MultiFab m_sig2, m_a1;
    bool m_cf = false;
    const MultiFab* m_crse = nullptr;
    Geometry m_cgeom;
    int m_ratio = 2;
    std::vector<Level> m_lev;
};

// nodal Laplacian div(sigma grad phi) = rhs  (MLNodeLaplacian + MLMG role)
class NodalMG {
public:
    NodalMG(const Geometry& g, LayoutP layout, const DomainBC& bc, const MGOpts& o);
    // sigma: cell centred, valid region used (ghost cells are filled internally, as MLNodeLaplacian does)
    void setSigma(const MultiFab& sig, int comp);
    MGStats solve(MultiFab& phi, const MultiFab& rhs, double rtol, double atol);
    int nlevels() const { return (int)m_lev.size(); }
    bool masked() const { return m_masked; }
    const MultiFab* dmask(int l) const { return m_lev[l].dmask(); }
    const MultiFab& sigma(int l) const { return m_lev[l].sig; }
    const Geometry& geom(int l) const { return m_lev[l].g; }
    // x_is_zero: x is to be taken as zero whatever it holds (the first smooth call on a correction); the call leaves x fully defined
    // leave_ghosts: the caller reads no ghost node of x afterwards (the fill behind the last sweep is skipped); x_filled: x comes straight from smooth()
    void smooth(int l, MultiFab& x, const MultiFab& rhs, bool x_is_zero = false, bool leave_ghosts = false);
    void residual(int l, MultiFab& r, MultiFab& x, const MultiFab& b, double* norm = nullptr, bool x_filled = false);   // norm: max norm of r (by the residual launch itself where it can)
    void vcycle(MGStats& st);
    // one V-cycle for the residual equation A e = r, zero initial guess; e is zero on Dirichlet nodes, its ghost nodes are filled.
    // Building block of the composite (multi-level) solver, amrns.hip.
    void vcycle_correction(MultiFab& e, const MultiFab& r, MGStats& st);
    void vcycle_correction_inplace(MGStats& st);      // r in res(0) (valid nodes), e in cor(0) (valid + 1 ghost layer)
    MultiFab& res(int l) { return m_lev[l].res; }
    MultiFab& cor(int l) { return m_lev[l].cor; }

private:
    struct Level {
        Geometry g;
        LayoutP layout;
        MultiFab sig;              // cell, 1 ghost
        MultiFab cor, res, rescor; // node, 1 ghost
        MultiFab tmp;              // Jacobi scratch
        bool agg = false;          // see CellMG::Level
        LayoutP dist;
        MultiFab tmp_d;
        bool slab = false;         // slab level, see CellMG::Level
        LayoutP virt;
        MultiFab vres;
        MultiFab xb;               // second buffer of the out-of-place fused Gauss-Seidel sweeps
        bool res_filled = false;   // the ghost nodes of `res` are current (filled once per V-cycle, not once per smooth call)
        MultiFab dm;               // Dirichlet node mask (defined only if the level has Dirichlet nodes, see NodalMG ctor)
        const MultiFab* dmask() const { return dm.defined() ? &dm : nullptr; }
    };
    int bicgstab(int l, MultiFab& sol, const MultiFab& rhs, double eps_rel, double eps_abs, int& niters);
    bool bottom_on_device();     // the coarsest level is solved by k_nodal_bottom (single-workgroup launch, no host synchronisation)
    void subtract_mean(int l, MultiFab& mf);
    void fillbc(int l, MultiFab& x, int kpar = -1, hipStream_t on = nullptr);   // kpar = 0 / 1: refresh the ghost nodes of the z-planes of that parity only
    Geometry m_g;
    DomainBC m_bc;
    MGOpts m_o;
    bool m_singular = true;
    bool m_csig = false;           // sigma is one constant on every level (constant density): the smoother skips the sigma planes
    double m_csig_val = 0.0;
    bool m_masked = false;         // some nodes are Dirichlet nodes (outflow faces / level boundary inside the domain)
    std::vector<Level> m_lev;
};

}  // namespace iamrx
