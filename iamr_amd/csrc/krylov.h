// iamr_amd/csrc/krylov.h -- BiCGStab of the multigrid bottom solvers with its scalars RESIDENT ON THE DEVICE (round 6).
//
// amrex::MLCGSolver::solve_bicgstab as CellMG::bicgstab / NodalMG::bicgstab restate it reads five scalars back per iteration (rho, rh.v,
// |s|, (t.t, t.s), |r|) and issues ~30 small launches: on the coarsest level of a regridded refined level (BASELINE configs C3 / C5: boxes
// that coarsen only twice, 10^5 ... 10^6 unknowns, ~13 iterations per V-cycle) the host waits for the device five times per iteration and the
// device for the host in between.  Here the dot products and norms stay where the reductions leave them (reduce_dots_dev,
// reduce_max_f_dev), alpha / beta / omega are formed by the kernels that use them (every thread divides the same two doubles: the same
// IEEE quotient the host formed), the convergence and break-down tests run in a one-thread control kernel behind each norm, and the host
// reads ONE status word per iteration.  Kernels of an iteration that has already ended find the status set and do nothing.
// Same operations on the same doubles in the same order as the host-driven loop: the iterates, the iteration count and the return code are
// those of CellMG::bicgstab / NodalMG::bicgstab (IAMRX_KRYLOV_DEVICE = 0 runs the host-driven loop; tests/test_gpu_krylov.py compares).
#pragma once
#include "operators.h"
#include "launch.h"

namespace iamrx {

// device scalars of one solve
enum { KS_RHO = 0, KS_RHO1, KS_ALPHA, KS_OMEGA, KS_RHTV, KS_TT, KS_TS, KS_NORM, KS_STATUS, KS_RET, KS_RNORM, KS_N };
// KS_STATUS: 0 running; 1 converged behind the half step; 2 converged behind the full step; 3 break-down (KS_RET = 1 ... 4)

#ifdef __HIPCC__
static __global__ void k_krylov_init(double* S, double rnorm0)
{
    for (int q = 0; q < KS_N; ++q) S[q] = 0.0;
    S[KS_RNORM] = rnorm0;
}
// The tests of the host-driven loop, in its order, behind |s| (stage 1) and behind |r| (stage 2).  The update kernels in front of a stage
// have already skipped their work where a test of that stage fails (they read the same scalars), so nothing is left to undo.
static __global__ void k_krylov_ctl(double* S, int stage, double rnorm0, double eps_rel, double eps_abs)
{
    if (S[KS_STATUS] != 0.0) return;
    if (stage == 1) {
        if (S[KS_RHO] == 0.0) { S[KS_STATUS] = 3.0; S[KS_RET] = 1.0; return; }
        if (S[KS_RHTV] == 0.0) { S[KS_STATUS] = 3.0; S[KS_RET] = 2.0; return; }
        S[KS_ALPHA] = S[KS_RHO] / S[KS_RHTV];
        const double rn = S[KS_NORM];
        S[KS_RNORM] = rn;
        if (rn < eps_rel * rnorm0 || rn < eps_abs) S[KS_STATUS] = 1.0;
        return;
    }
    if (S[KS_TT] == 0.0) { S[KS_STATUS] = 3.0; S[KS_RET] = 3.0; return; }
    const double omega = S[KS_TS] / S[KS_TT];
    S[KS_OMEGA] = omega;
    const double rn = S[KS_NORM];
    S[KS_RNORM] = rn;
    if (rn < eps_rel * rnorm0 || rn < eps_abs) { S[KS_STATUS] = 2.0; return; }
    if (omega == 0.0) { S[KS_STATUS] = 3.0; S[KS_RET] = 4.0; return; }
    S[KS_RHO1] = S[KS_RHO];
}
#endif

// One solve.  apply(out, in): fill the ghost points of `in` (a ghosted work array whose valid points hold the vector), then out = A in on
// the valid points, masked as the operator needs.  sol: zero on entry; r: the initial residual (overwritten); rh: a copy of r; ph, sh: ghosted
// work arrays, zero on entry (p and s live on their valid points); v, t: work arrays.  Single rank (or replicated layout) only: the status
// word of iteration n is read while iteration n + 1 is already queued, so the host never holds the device up; an iteration queued behind
// the end of the solve does nothing.  Returns the MLCGSolver code (0, 1 ... 4; the caller adds 8); niters, rnorm as the host-driven loop.
template <class Apply>
int bicgstab_device(const Layout& lay, const IndexType& type, int nc, const Geometry& g, MultiFab& sol, MultiFab& r, const MultiFab& rh, MultiFab& ph, MultiFab& sh,
                    MultiFab& v, MultiFab& t, double rnorm0, double eps_rel, double eps_abs, int maxiter, Apply apply, int& niters, double& rnorm)
{
    auto& ctx = Context::get();
    static double* S = nullptr;
    static double* hS = nullptr;       // two pinned copies, alternating
    static hipEvent_t ev[2];
    if (!S) {
        IAMRX_HIP_CHECK(hipMalloc(&S, KS_N * sizeof(double)));
        IAMRX_HIP_CHECK(hipHostMalloc(&hS, 2 * KS_N * sizeof(double)));
        for (int q = 0; q < 2; ++q) IAMRX_HIP_CHECK(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
    }
    hipLaunchKernelGGL(k_krylov_init, dim3(1), dim3(1), 0, ctx.stream, S, rnorm0);
    const FabD *solt = sol.d_tab, *rt = r.d_tab, *pht = ph.d_tab, *sht = sh.d_tab, *vt = v.d_tab, *tt = t.d_tab;
    const double* Sc = S;
    auto status_of = [&](int it) {       // blocks until iteration `it` has run
        IAMRX_HIP_CHECK(hipEventSynchronize(ev[it & 1]));
        return (int)hS[(it & 1) * KS_N + KS_STATUS];
    };
    int nit = 1, ended = 0;              // ended: the iteration whose status word came back non-zero
    for (; nit <= maxiter; ++nit) {
        { const MultiFab* xs[1] = {&rh}; const MultiFab* ys[1] = {&r}; reduce_dots_dev(1, xs, ys, 0, nc, g, S + KS_RHO); }
        // p = r (first iteration) or r + beta (p - omega v), beta = (rho / rho_1) (alpha / omega)
        const bool first = nit == 1;
        for_each(lay, type, 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
            if (Sc[KS_STATUS] != 0.0 || Sc[KS_RHO] == 0.0) return;
            if (first) { for (int n = 0; n < nc; ++n) pht[f](i, j, k, n) = rt[f](i, j, k, n); return; }
            const double beta = (Sc[KS_RHO] / Sc[KS_RHO1]) * (Sc[KS_ALPHA] / Sc[KS_OMEGA]), momega = -Sc[KS_OMEGA];
            for (int n = 0; n < nc; ++n) {
                const double t1 = 1.0 * pht[f](i, j, k, n) + momega * vt[f](i, j, k, n);
                pht[f](i, j, k, n) = 1.0 * rt[f](i, j, k, n) + beta * t1;
            }
        });
        apply(v, ph);
        { const MultiFab* xs[1] = {&rh}; const MultiFab* ys[1] = {&v}; reduce_dots_dev(1, xs, ys, 0, nc, g, S + KS_RHTV); }
        // alpha = rho / (rh.v); sol += alpha p; s = r - alpha v; |s|
        reduce_max_f_dev<1>(lay, type, 0, [=] __device__(int i, int j, int k, int f, double* m) {
            if (Sc[KS_STATUS] != 0.0 || Sc[KS_RHO] == 0.0 || Sc[KS_RHTV] == 0.0) return;
            const double alpha = Sc[KS_RHO] / Sc[KS_RHTV], malpha = -alpha;
            for (int n = 0; n < nc; ++n) {
                solt[f](i, j, k, n) = 1.0 * solt[f](i, j, k, n) + alpha * pht[f](i, j, k, n);
                const double s = 1.0 * rt[f](i, j, k, n) + malpha * vt[f](i, j, k, n);
                sht[f](i, j, k, n) = s;
                const double a = fabs(s);
                m[0] = a > m[0] ? a : m[0];
            }
        }, S + KS_NORM);
        hipLaunchKernelGGL(k_krylov_ctl, dim3(1), dim3(1), 0, ctx.stream, S, 1, rnorm0, eps_rel, eps_abs);
        apply(t, sh);
        { const MultiFab* xs[2] = {&t, &t}; const MultiFab* ys[2] = {&t, &sh}; reduce_dots_dev(2, xs, ys, 0, nc, g, S + KS_TT); }
        // omega = (t.s) / (t.t); sol += omega s; r = s - omega t; |r|
        reduce_max_f_dev<1>(lay, type, 0, [=] __device__(int i, int j, int k, int f, double* m) {
            if (Sc[KS_STATUS] != 0.0 || Sc[KS_TT] == 0.0) return;
            const double omega = Sc[KS_TS] / Sc[KS_TT], momega = -omega;
            for (int n = 0; n < nc; ++n) {
                const double s = sht[f](i, j, k, n);
                solt[f](i, j, k, n) = 1.0 * solt[f](i, j, k, n) + omega * s;
                const double rn = 1.0 * s + momega * tt[f](i, j, k, n);
                rt[f](i, j, k, n) = rn;
                const double a = fabs(rn);
                m[0] = a > m[0] ? a : m[0];
            }
        }, S + KS_NORM);
        hipLaunchKernelGGL(k_krylov_ctl, dim3(1), dim3(1), 0, ctx.stream, S, 2, rnorm0, eps_rel, eps_abs);
        IAMRX_HIP_CHECK(hipMemcpyAsync(hS + (nit & 1) * KS_N, S, KS_N * sizeof(double), hipMemcpyDeviceToHost, ctx.stream));
        IAMRX_HIP_CHECK(hipEventRecord(ev[nit & 1], ctx.stream));
        if (nit > 1 && status_of(nit - 1) != 0) { ended = nit - 1; break; }
    }
    if (!ended) {
        const int last = nit > maxiter ? maxiter : nit;
        if (status_of(last) != 0) ended = last;
    }
    const double* h = hS + ((ended ? ended : maxiter) & 1) * KS_N;
    rnorm = h[KS_RNORM];
    niters = ended ? ended : maxiter + 1;   // the loop counter of the host-driven loop where it stops
    return (int)h[KS_STATUS] == 3 ? (int)h[KS_RET] : 0;
}

}  // namespace iamrx
