// iamr_amd/csrc/k_godunov.hip -- Godunov (PLM + corner-transport-upwind) kernels for gfx950.
//
// Role: AMReX-Hydro Godunov::ExtrapVelToFaces (reference call site Source/NavierStokesBase.cpp:4487-4491)
// and HydroUtils::ComputeFluxesOnBoxFromState / ComputeDivergence / ComputeConvectiveTerm
// (Source/NavierStokesBase.cpp:4701-4842), SURVEY a3 / a8.
//
// MI355X-first structure (NOT the reference's ~20 scratch arrays per box):
//   pass 1  k_trace   : per face of each direction (transverse grown by 1): 4th-order limited slopes are
//                       evaluated in registers, the two traced states are upwinded at once ->
//                       advective velocity (predict mode) + single-valued transverse states.
//   pass 2  k_final   : per face: the four corner-coupled transverse states per transverse direction are
//                       rebuilt in registers from pass-1 data (re-evaluating slopes instead of storing
//                       lo/hi arrays), transverse terms + forcing + BCs + final upwinding.
//   pass 3  k_aofs    : per cell: area-weighted fluxes, -div, convective correction, aofs = -update.
// Only 3+3*ncomp face arrays are materialised in HBM between the passes.
// All index shifts are done with linear strides so one code path serves the three directions.
#include "kernels.h"
#include "launch.h"

namespace iamrx {

#define SMALL_VEL 1.e-8

struct GodBC {
    int dlo[3], dhi[3];
    int per[3];
    BCRec bc[5];
};

__device__ __forceinline__ double lim2(double dlft, double drgt)
{
    const double dcen = 0.5 * (dlft + drgt);
    const double dsgn = copysign(1.0, dcen);
    const double dlim = (dlft * drgt >= 0.0) ? 2.0 * fmin(fabs(dlft), fabs(drgt)) : 0.0;
    return dsgn * fmin(dlim, fabs(dcen));
}

// amrex_calc_{x,y,z}slope_extdir, order 4.  q points at the cell, s = stride in the slope direction,
// i = cell index in that direction.
__device__ __forceinline__ double slope4(const double* __restrict__ q, long s, bool edlo, bool edhi, int i, int domlo, int domhi)
{
    const double qi = q[0], qm = q[-s], qp = q[s], qmm = q[-2 * s], qpp = q[2 * s];
    double dfm = lim2(qm - qmm, qi - qm);
    double dfp = lim2(qp - qi, qpp - qp);
    double dlft = qi - qm, drgt = qp - qi;
    const double dcen = 0.5 * (dlft + drgt);
    double dsgn = copysign(1.0, dcen);
    double dlim = (dlft * drgt >= 0.0) ? 2.0 * fmin(fabs(dlft), fabs(drgt)) : 0.0;
    double dtemp = 4.0 / 3.0 * dcen - 1.0 / 6.0 * (dfp + dfm);
    if (edlo && i == domlo) {
        dtemp = -16. / 15. * qm + .5 * qi + 2. / 3. * qp - 0.1 * qpp;
        dlft = 2. * (qi - qm); drgt = 2. * (qp - qi);
        dlim = (dlft * drgt >= 0.0) ? fmin(fabs(dlft), fabs(drgt)) : 0.0;
        dsgn = copysign(1.0, dtemp);
    } else if (edlo && i == domlo + 1) {
        dfm = -16. / 15. * qmm + .5 * qm + 2. / 3. * qi - 0.1 * qp;
        const double l = 2. * (qm - qmm), r = 2. * (qi - qm);
        const double dlimsh = (l * r >= 0.0) ? fmin(fabs(l), fabs(r)) : 0.0;
        const double dsgnsh = copysign(1.0, dfm);
        dfm = dsgnsh * fmin(dlimsh, fabs(dfm));
        dtemp = 4.0 / 3.0 * dcen - 1.0 / 6.0 * (dfp + dfm);
    }
    if (edhi && i == domhi) {
        dtemp = 16. / 15. * qp - .5 * qi - 2. / 3. * qm + 0.1 * qmm;
        dlft = 2. * (qi - qm); drgt = 2. * (qp - qi);
        dlim = (dlft * drgt >= 0.0) ? fmin(fabs(dlft), fabs(drgt)) : 0.0;
        dsgn = copysign(1.0, dtemp);
    } else if (edhi && i == domhi - 1) {
        dfp = 16. / 15. * qpp - .5 * qp - 2. / 3. * qi + 0.1 * qm;
        const double l = 2. * (qp - qi), r = 2. * (qpp - qp);
        const double dlimsh = (l * r >= 0.0) ? fmin(fabs(l), fabs(r)) : 0.0;
        const double dsgnsh = copysign(1.0, dfp);
        dfp = dsgnsh * fmin(dlimsh, fabs(dfp));
        dtemp = 4.0 / 3.0 * dcen - 1.0 / 6.0 * (dfp + dfm);
    }
    return dsgn * fmin(dlim, fabs(dtemp));
}

__device__ __forceinline__ bool ed_or_ho(int b) { return b == bc_ext_dir || b == bc_hoextrap; }

// SetTransTerm{X,Y,Z}BCs.  qc points at the state value of the cell on the HIGH side of the face
// (cell index f in direction d), s = stride in d, f = face index.
__device__ __forceinline__ void trans_bc(const double* __restrict__ qc, long s, int f, bool normal_vel, double& lo, double& hi,
                                         int bclo, int bchi, int domlo, int domhi)
{
    if (f <= domlo) {
        if (bclo == bc_ext_dir) { lo = qc[(long)(domlo - 1 - f) * s]; if (normal_vel) hi = lo; }
        else if (bclo == bc_foextrap || bclo == bc_hoextrap || bclo == bc_reflect_even) lo = hi;
        else if (bclo == bc_reflect_odd) { hi = 0.; lo = 0.; }
    } else if (f > domhi) {
        if (bchi == bc_ext_dir) { hi = qc[(long)(domhi + 1 - f) * s]; if (normal_vel) lo = hi; }
        else if (bchi == bc_foextrap || bchi == bc_hoextrap || bchi == bc_reflect_even) hi = lo;
        else if (bchi == bc_reflect_odd) { lo = 0.; hi = 0.; }
    }
}

// Set{X,Y,Z}EdgeBCs
__device__ __forceinline__ void edge_bc(const double* __restrict__ qc, long s, int f, bool normal_vel, double& lo, double& hi,
                                        int bclo, int bchi, int domlo, int domhi)
{
    if (f <= domlo) {
        if (bclo == bc_ext_dir) { lo = qc[(long)(domlo - 1 - f) * s]; if (normal_vel) hi = lo; }
        else if (bclo == bc_foextrap || bclo == bc_hoextrap || bclo == bc_reflect_even) {
            if (normal_vel && bclo != bc_reflect_even) hi = fmin(hi, 0.);
            lo = hi;
        } else if (bclo == bc_reflect_odd) { hi = 0.; lo = 0.; }
    } else if (f > domhi) {
        if (bchi == bc_ext_dir) { hi = qc[(long)(domhi + 1 - f) * s]; if (normal_vel) lo = hi; }
        else if (bchi == bc_foextrap || bchi == bc_hoextrap || bchi == bc_reflect_even) {
            if (normal_vel && bchi != bc_reflect_even) lo = fmax(lo, 0.);
            hi = lo;
        } else if (bchi == bc_reflect_odd) { lo = 0.; hi = 0.; }
    }
}

// traced states on face f of direction d for component n: lo from cell f-1, hi from cell f.
// PRED: trace velocity = cell-centred vcc(cell, d); else the face's own mac velocity `um`.
template <bool PRED>
__device__ __forceinline__ void trace_lohi(const double* __restrict__ qn /*state comp n at cell f*/, const double* __restrict__ vd /*vcc comp d at cell f (PRED)*/,
                                           long s, double um, double dtdx, bool edlo, bool edhi, int f, int domlo, int domhi,
                                           double& lo, double& hi)
{
    const double slh = slope4(qn, s, edlo, edhi, f, domlo, domhi);
    const double sll = slope4(qn - s, s, edlo, edhi, f - 1, domlo, domhi);
    if (PRED) {
        hi = qn[0] + 0.5 * (-1.0 - vd[0] * dtdx) * slh;
        lo = qn[-s] + 0.5 * (1.0 - vd[-s] * dtdx) * sll;
    } else {
        hi = qn[0] + 0.5 * (-1.0 - um * dtdx) * slh;
        lo = qn[-s] + 0.5 * (1.0 - um * dtdx) * sll;
    }
}

struct GodParams {
    double dt;
    double dx[3];
    int ncomp;
    int is_velocity;
    int fit;            // use_forces_in_trans
    int has_force;
    int has_divu;
    int iconserv[5];
    GodBC bc;
};

// -------------------------------------------------------------------------------- pass 1
// grid over faces of direction D, transverse directions grown by 1.
// PRED: writes ad[D] (1 comp) and e0[D] (ncomp comps, upwinded with ad).  ADV: mac given, writes e0[D].
template <bool PRED>
__global__ void __launch_bounds__(256) k_trace(Tiling t, const BoxD* __restrict__ boxes, int D,
    const FabD* __restrict__ qt, const FabD* __restrict__ ft, const FabD* __restrict__ mact /*ADV: umac[D]; PRED: out ad[D]*/,
    const FabD* __restrict__ e0t, GodParams P)
{
    const int fab = blockIdx.y;
    BoxD b = boxes[fab];
    for (int e = 0; e < 3; ++e) { if (e == D) b.hi[e] += 1; else { b.lo[e] -= 1; b.hi[e] += 1; } }
    int i, j, k0, k1;
    if (!tile_ijk(t, b, i, j, k0, k1)) return;
    const FabD q = qt[fab], e0 = e0t[fab], mac = mact[fab];
    FabD frc; if (P.has_force) frc = ft[fab];
    const long qs[3] = {1, q.n[0], (long)q.n[0] * q.n[1]};
    const long s = qs[D];
    const double dtdx = P.dt / P.dx[D];
    const int domlo = P.bc.dlo[D], domhi = P.bc.dhi[D];
    const bool nonper = !P.bc.per[D];
    for (int k = k0; k <= k1; ++k) {
        const int idx[3] = {i, j, k};
        const int f = idx[D];
        const long qo = q.off(i, j, k);
        double um = 0.0;
        if (!PRED) um = mac(i, j, k, 0);
        double lo[5], hi[5];
        for (int n = 0; n < P.ncomp; ++n) {
            const int bl = P.bc.bc[n].lo[D], bh = P.bc.bc[n].hi[D];
            const bool edlo = nonper && ed_or_ho(bl), edhi = nonper && ed_or_ho(bh);
            trace_lohi<PRED>(q.p + qo + q.cs * n, q.p + qo + q.cs * D, s, um, dtdx, edlo, edhi, f, domlo, domhi, lo[n], hi[n]);
            if (P.fit && P.has_force) {
                const long fo = frc.off(i, j, k);
                const long fs = D == 0 ? 1 : (D == 1 ? frc.n[0] : (long)frc.n[0] * frc.n[1]);
                lo[n] += 0.5 * P.dt * frc.p[fo - fs + frc.cs * n];
                hi[n] += 0.5 * P.dt * frc.p[fo + frc.cs * n];
            }
            if (nonper) trans_bc(q.p + qo + q.cs * n, s, f, P.is_velocity && n == D, lo[n], hi[n], bl, bh, domlo, domhi);
        }
        double uad;
        if (PRED) {
            const double l = lo[D], h = hi[D];
            const double st = ((l + h) >= 0.) ? l : h;
            const bool ltm = ((l <= 0. && h >= 0.) || (fabs(l + h) < SMALL_VEL));
            uad = ltm ? 0. : st;
            mac(i, j, k, 0) = uad;
        } else uad = um;
        const double fu = (fabs(uad) < SMALL_VEL) ? 0.0 : 1.0;
        for (int n = 0; n < P.ncomp; ++n) {
            const double st = (uad >= 0.) ? lo[n] : hi[n];
            e0(i, j, k, n) = fu * st + (1.0 - fu) * 0.5 * (hi[n] + lo[n]);
        }
    }
}

// corner-coupled, upwinded state on the T-face at index position (pointer offsets already applied):
//   qn  : state comp n at the cell on the high side of the T-face
//   lo/hi traced along T, corrected with the O-derivative built from mac[O] and e0[O], upwinded with mac[T]
template <bool PRED>
__device__ __forceinline__ double corner_state(const double* __restrict__ qn, const double* __restrict__ vT, long sT, int fT,
    double macT_f, const double* __restrict__ macO, long mOsT, long mOsO,
    const double* __restrict__ eO, long eOsT, long eOsO,
    const double* __restrict__ frcn, long fsT, double dtdxT, double c_o /* dt/(6 dxO) or dt/(3 dxO) */, double dt3, double dxO,
    bool conserv, const double* __restrict__ divu, long dsT, bool has_divu,
    bool fit, double hdt, bool nonperT, bool normal_vel, int bl, int bh, int domlo, int domhi)
{
    const bool edlo = nonperT && ed_or_ho(bl), edhi = nonperT && ed_or_ho(bh);
    double l, h;
    trace_lohi<PRED>(qn, vT, sT, macT_f, dtdxT, edlo, edhi, fT, domlo, domhi, l, h);
    if (fit && frcn) { l += hdt * frcn[-fsT]; h += hdt * frcn[0]; }
    if (nonperT) trans_bc(qn, sT, fT, normal_vel, l, h, bl, bh, domlo, domhi);   // BCs of the traced states (pass 1 order)
    // mac[O] / e0[O] at the low-side cell (cm = f - eT) and at the high-side cell (f), O-faces c and c+eO
    const double mo_cm = macO[-mOsT], mo_cmo = macO[-mOsT + mOsO], mo_f = macO[0], mo_fo = macO[mOsO];
    const double eo_cm = eO[-eOsT], eo_cmo = eO[-eOsT + eOsO], eo_f = eO[0], eo_fo = eO[eOsO];
    if (conserv) {
        const double dvl = has_divu ? divu[-dsT] : 0.0, dvh = has_divu ? divu[0] : 0.0;
        l = l - c_o * (eo_cmo * mo_cmo - eo_cm * mo_cm) + dt3 * qn[-sT] * ((mo_cmo - mo_cm) / dxO - 0.5 * dvl);
        h = h - c_o * (eo_fo * mo_fo - eo_f * mo_f) + dt3 * qn[0] * ((mo_fo - mo_f) / dxO - 0.5 * dvh);
    } else {
        l = l - c_o * (mo_cmo + mo_cm) * (eo_cmo - eo_cm);
        h = h - c_o * (mo_fo + mo_f) * (eo_fo - eo_f);
    }
    if (nonperT) trans_bc(qn, sT, fT, normal_vel, l, h, bl, bh, domlo, domhi);
    const double st = (macT_f >= 0.) ? l : h;
    const double fu = (fabs(macT_f) < SMALL_VEL) ? 0.0 : 1.0;
    return fu * st + (1.0 - fu) * 0.5 * (h + l);
}

// -------------------------------------------------------------------------------- pass 2
// grid over the valid faces of direction D.  PRED: only component n = D, output umac[D];
// ADV: all components, output final edge states edge[D](ncomp).
template <bool PRED>
__global__ void __launch_bounds__(256) k_final(Tiling t, const BoxD* __restrict__ boxes, int D,
    const FabD* __restrict__ qt, const FabD* __restrict__ ft, const FabD* __restrict__ divut,
    const FabD* __restrict__ m0t, const FabD* __restrict__ m1t, const FabD* __restrict__ m2t,
    const FabD* __restrict__ e0t, const FabD* __restrict__ e1t, const FabD* __restrict__ e2t,
    const FabD* __restrict__ outt, GodParams P)
{
    const int fab = blockIdx.y;
    BoxD b = boxes[fab];
    b.hi[D] += 1;
    int i, j, k0, k1;
    if (!tile_ijk(t, b, i, j, k0, k1)) return;
    const FabD q = qt[fab], out = outt[fab];
    const FabD mac[3] = {m0t[fab], m1t[fab], m2t[fab]};
    const FabD e0[3] = {e0t[fab], e1t[fab], e2t[fab]};
    FabD frc; if (P.has_force) frc = ft[fab];
    FabD dv; if (P.has_divu) dv = divut[fab];
    const long qs[3] = {1, q.n[0], (long)q.n[0] * q.n[1]};
    long fs[3] = {0, 0, 0}, ds[3] = {0, 0, 0};
    if (P.has_force) { fs[0] = 1; fs[1] = frc.n[0]; fs[2] = (long)frc.n[0] * frc.n[1]; }
    if (P.has_divu) { ds[0] = 1; ds[1] = dv.n[0]; ds[2] = (long)dv.n[0] * dv.n[1]; }
    const double hdt = 0.5 * P.dt;
    const double dtdxD = P.dt / P.dx[D];
    const int nbeg = PRED ? D : 0, nend = PRED ? D + 1 : P.ncomp;
    for (int k = k0; k <= k1; ++k) {
        const int idx[3] = {i, j, k};
        const int f = idx[D];
        const long qo = q.off(i, j, k);
        const long fo = P.has_force ? frc.off(i, j, k) : 0;
        const long dvo = P.has_divu ? dv.off(i, j, k) : 0;
        const double umD = mac[D](i, j, k, 0);
        for (int n = nbeg; n < nend; ++n) {
            const double* qn = q.p + qo + q.cs * n;
            const double* frcn = P.has_force ? frc.p + fo + frc.cs * n : nullptr;
            const bool conserv = !PRED && P.iconserv[n];
            // own traced states along D
            double stl, sth;
            {
                const int bl = P.bc.bc[n].lo[D], bh = P.bc.bc[n].hi[D];
                const bool nonper = !P.bc.per[D];
                const bool edlo = nonper && ed_or_ho(bl), edhi = nonper && ed_or_ho(bh);
                trace_lohi<PRED>(qn, q.p + qo + q.cs * D, qs[D], umD, dtdxD, edlo, edhi, f, P.bc.dlo[D], P.bc.dhi[D], stl, sth);
                if (P.fit && P.has_force) { stl += hdt * frcn[-fs[D]]; sth += hdt * frcn[0]; }
                if (nonper) trans_bc(qn, qs[D], f, P.is_velocity && n == D, stl, sth, bl, bh, P.bc.dlo[D], P.bc.dhi[D]);
            }
            double Tl[3][2], Th[3][2];   // corner states on the T-faces of the low-side cell (cm) and the high-side cell (f): [T][0]=face c, [1]=face c+eT
            for (int T = 0; T < 3; ++T) {
                if (T == D) continue;
                const int O = 3 - D - T;
                const FabD& mT = mac[T]; const FabD& mO = mac[O]; const FabD& eO = e0[O];
                const long mTs[3] = {1, mT.n[0], (long)mT.n[0] * mT.n[1]};
                const long mOs[3] = {1, mO.n[0], (long)mO.n[0] * mO.n[1]};
                const long eOs[3] = {1, eO.n[0], (long)eO.n[0] * eO.n[1]};
                const long mTo = mT.off(i, j, k), mOo = mO.off(i, j, k), eOo = eO.off(i, j, k) + eO.cs * n;
                const double c_o = conserv ? P.dt / (3.0 * P.dx[O]) : P.dt / (6.0 * P.dx[O]);
                const double dt3 = P.dt / 3.0;
                const double dtdxT = P.dt / P.dx[T];
                const int bl = P.bc.bc[n].lo[T], bh = P.bc.bc[n].hi[T];
                const bool nonperT = !P.bc.per[T];
                const bool nvel = P.is_velocity && n == T;
                for (int side = 0; side < 2; ++side) {          // 0: low-side cell cm = f - eD ; 1: high-side cell f
                    const long shD_q = side ? 0 : -qs[D];
                    for (int up = 0; up < 2; ++up) {            // T-face index c (0) or c+1 (1) of that cell
                        const long oq = shD_q + up * qs[T];
                        const long om = (side ? 0 : -mTs[D]) + up * mTs[T];
                        const long oO = (side ? 0 : -mOs[D]) + up * mOs[T];
                        const long oe = (side ? 0 : -eOs[D]) + up * eOs[T];
                        const long of = (side ? 0 : -fs[D]) + up * fs[T];
                        const long od = (side ? 0 : -ds[D]) + up * ds[T];
                        const double v = corner_state<PRED>(qn + oq, q.p + qo + q.cs * T + oq, qs[T], idx[T] + up,
                            mT.p[mTo + om], mO.p + mOo + oO, mOs[T], mOs[O], eO.p + eOo + oe, eOs[T], eOs[O],
                            P.has_force ? frcn + of : nullptr, fs[T], dtdxT, c_o, dt3, P.dx[O], conserv,
                            P.has_divu ? dv.p + dvo + od : nullptr, ds[T], P.has_divu != 0,
                            P.fit != 0, hdt, nonperT, nvel, bl, bh, P.bc.dlo[T], P.bc.dhi[T]);
                        if (side == 0) Tl[T][up] = v; else Th[T][up] = v;
                    }
                }
            }
            // transverse terms, ascending transverse direction
            if (conserv) {
                for (int T = 0; T < 3; ++T) {
                    if (T == D) continue;
                    const FabD& mT = mac[T];
                    const long mTs[3] = {1, mT.n[0], (long)mT.n[0] * mT.n[1]};
                    const long mTo = mT.off(i, j, k);
                    const double c = 0.5 * P.dt / P.dx[T];
                    stl += -c * (Tl[T][1] * mT.p[mTo - mTs[D] + mTs[T]] - Tl[T][0] * mT.p[mTo - mTs[D]]);
                    sth += -c * (Th[T][1] * mT.p[mTo + mTs[T]] - Th[T][0] * mT.p[mTo]);
                }
                for (int T = 0; T < 3; ++T) {
                    if (T == D) continue;
                    const FabD& mT = mac[T];
                    const long mTs[3] = {1, mT.n[0], (long)mT.n[0] * mT.n[1]};
                    const long mTo = mT.off(i, j, k);
                    const double c = 0.5 * P.dt / P.dx[T];
                    stl += c * qn[-qs[D]] * (mT.p[mTo - mTs[D] + mTs[T]] - mT.p[mTo - mTs[D]]);
                    sth += c * qn[0] * (mT.p[mTo + mTs[T]] - mT.p[mTo]);
                }
                if (P.has_divu) { stl -= 0.5 * P.dt * qn[-qs[D]] * dv.p[dvo - ds[D]]; sth -= 0.5 * P.dt * qn[0] * dv.p[dvo]; }
            } else {
                for (int T = 0; T < 3; ++T) {
                    if (T == D) continue;
                    const FabD& mT = mac[T];
                    const long mTs[3] = {1, mT.n[0], (long)mT.n[0] * mT.n[1]};
                    const long mTo = mT.off(i, j, k);
                    const double c = 0.25 * P.dt / P.dx[T];
                    stl -= c * (mT.p[mTo - mTs[D] + mTs[T]] + mT.p[mTo - mTs[D]]) * (Tl[T][1] - Tl[T][0]);
                    sth -= c * (mT.p[mTo + mTs[T]] + mT.p[mTo]) * (Th[T][1] - Th[T][0]);
                }
            }
            if (!P.fit && P.has_force) { stl += hdt * frcn[-fs[D]]; sth += hdt * frcn[0]; }
            if (!P.bc.per[D]) edge_bc(qn, qs[D], f, P.is_velocity && n == D, stl, sth, P.bc.bc[n].lo[D], P.bc.bc[n].hi[D], P.bc.dlo[D], P.bc.dhi[D]);
            if (PRED) {
                const double st = ((stl + sth) >= 0.) ? stl : sth;
                const bool ltm = ((stl <= 0. && sth >= 0.) || (fabs(stl + sth) < SMALL_VEL));
                out(i, j, k, 0) = ltm ? 0. : st;
            } else {
                double temp = (umD >= 0.) ? stl : sth;
                temp = (fabs(umD) < SMALL_VEL) ? 0.5 * (stl + sth) : temp;
                out(i, j, k, n) = temp;
            }
        }
    }
}

static GodParams make_params(const Geometry& g, double dt, int ncomp, const BCRec* bc, const int* iconserv, bool is_vel, bool fit,
                             bool has_force, bool has_divu)
{
    GodParams P;
    P.dt = dt; P.ncomp = ncomp; P.is_velocity = is_vel; P.fit = fit; P.has_force = has_force; P.has_divu = has_divu;
    for (int d = 0; d < 3; ++d) { P.dx[d] = g.dx[d]; P.bc.dlo[d] = g.domain.lo[d]; P.bc.dhi[d] = g.domain.hi[d]; P.bc.per[d] = g.periodic[d]; }
    for (int n = 0; n < 5; ++n) {
        P.iconserv[n] = (iconserv && n < ncomp) ? iconserv[n] : 0;
        for (int d = 0; d < 3; ++d) { P.bc.bc[n].lo[d] = (bc && n < ncomp) ? bc[n].lo[d] : 0; P.bc.bc[n].hi[d] = (bc && n < ncomp) ? bc[n].hi[d] : 0; }
    }
    return P;
}

static Tiling face_tiling(const Layout& l, int D, int gt, int tz)
{
    int ml[3];
    for (int e = 0; e < 3; ++e) ml[e] = l.max_len[e] + (e == D ? 1 : 2 * gt);
    return make_tiling(ml, l.nlocal(), tz);
}

void godunov_extrap_vel_to_faces(const Geometry& g, const MultiFab& vel, const MultiFab* force, MultiFab* const umac[3],
                                 double dt, const BCRec* bc, bool use_forces_in_trans)
{
    if (vel.nlocal() == 0) return;
    IAMRX_ASSERT(vel.ngrow >= 3 && vel.ncomp >= 3);
    IAMRX_ASSERT(!force || force->ngrow >= 1);
    auto& ctx = Context::get();
    const Layout& l = *vel.layout;
    MultiFab ad[3], e0[3];
    for (int d = 0; d < 3; ++d) { ad[d].define(vel.layout, face_type(d), 1, 1); e0[d].define(vel.layout, face_type(d), 3, 1); }
    GodParams P = make_params(g, dt, 3, bc, nullptr, true, use_forces_in_trans, force != nullptr, false);
    for (int d = 0; d < 3; ++d) {
        Tiling t = face_tiling(l, d, 1, 4);
        hipLaunchKernelGGL((k_trace<true>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, d, vel.d_tab,
                           force ? force->d_tab : nullptr, ad[d].d_tab, e0[d].d_tab, P);
    }
    for (int d = 0; d < 3; ++d) {
        Tiling t = face_tiling(l, d, 0, 4);
        hipLaunchKernelGGL((k_final<true>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, d, vel.d_tab,
                           force ? force->d_tab : nullptr, nullptr, ad[0].d_tab, ad[1].d_tab, ad[2].d_tab,
                           e0[0].d_tab, e0[1].d_tab, e0[2].d_tab, umac[d]->d_tab, P);
    }
}

// -------------------------------------------------------------------------------- pass 3
__global__ void __launch_bounds__(256) k_aofs(Tiling t, const BoxD* __restrict__ boxes, const FabD* __restrict__ aofst, int acomp,
    const FabD* __restrict__ ext, const FabD* __restrict__ eyt, const FabD* __restrict__ ezt,
    const FabD* __restrict__ uxt, const FabD* __restrict__ uyt, const FabD* __restrict__ uzt,
    const FabD* __restrict__ fxt, const FabD* __restrict__ fyt, const FabD* __restrict__ fzt, GodParams P)
{
    const int fab = blockIdx.y;
    int i, j, k0, k1;
    if (!tile_ijk(t, boxes[fab], i, j, k0, k1)) return;
    const FabD aofs = aofst[fab], ex = ext[fab], ey = eyt[fab], ez = ezt[fab], ux = uxt[fab], uy = uyt[fab], uz = uzt[fab];
    const bool store_flux = fxt != nullptr;
    const double ax = P.dx[1] * P.dx[2], ay = P.dx[2] * P.dx[0], az = P.dx[0] * P.dx[1];
    const double qvol = 1.0 / (P.dx[0] * P.dx[1] * P.dx[2]);
    for (int k = k0; k <= k1; ++k) {
        const double uxl = ux(i, j, k), uxh = ux(i + 1, j, k), uyl = uy(i, j, k), uyh = uy(i, j + 1, k), uzl = uz(i, j, k), uzh = uz(i, j, k + 1);
        const double divum = 1.0 * ((uxh - uxl) / P.dx[0] + (uyh - uyl) / P.dx[1] + (uzh - uzl) / P.dx[2]);
        for (int n = 0; n < P.ncomp; ++n) {
            const double exl = ex(i, j, k, n), exh = ex(i + 1, j, k, n), eyl = ey(i, j, k, n), eyh = ey(i, j + 1, k, n), ezl = ez(i, j, k, n), ezh = ez(i, j, k + 1, n);
            const double fxl = exl * uxl * ax, fxh = exh * uxh * ax, fyl = eyl * uyl * ay, fyh = eyh * uyh * ay, fzl = ezl * uzl * az, fzh = ezh * uzh * az;
            if (store_flux) {
                fxt[fab](i, j, k, n) = fxl; fyt[fab](i, j, k, n) = fyl; fzt[fab](i, j, k, n) = fzl;
                // high faces on the box boundary are not owned by any other cell of this fab
                const BoxD bb = boxes[fab];
                if (i == bb.hi[0]) fxt[fab](i + 1, j, k, n) = fxh;
                if (j == bb.hi[1]) fyt[fab](i, j + 1, k, n) = fyh;
                if (k == bb.hi[2]) fzt[fab](i, j, k + 1, n) = fzh;
            }
            double upd = -1.0 * qvol * ((fxh - fxl) + (fyh - fyl) + (fzh - fzl));
            if (!P.iconserv[n]) {
                double qavg = exl + exh + eyl + eyh + ezl + ezh;
                qavg *= 1.0 / 6.0;
                upd += qavg * divum;
            }
            aofs(i, j, k, acomp + n) = -upd;
        }
    }
}

void godunov_compute_aofs(const Geometry& g, MultiFab& aofs, int acomp, const MultiFab& S, int ncomp, const MultiFab* force,
                          const MultiFab* divu, MultiFab* const umac[3], const int* iconserv, double dt, const BCRec* bc,
                          bool is_velocity, bool use_forces_in_trans, MultiFab* const edge_out[3], MultiFab* const flux_out[3])
{
    if (S.nlocal() == 0) return;
    IAMRX_ASSERT(S.ngrow >= 3 && ncomp <= 5 && S.ncomp >= ncomp);
    IAMRX_ASSERT(umac[0]->ngrow >= 1);
    auto& ctx = Context::get();
    const Layout& l = *S.layout;
    MultiFab e0[3], edge[3];
    MultiFab* ed[3];
    for (int d = 0; d < 3; ++d) {
        e0[d].define(S.layout, face_type(d), ncomp, 1);
        if (edge_out && edge_out[d]) ed[d] = edge_out[d];
        else { edge[d].define(S.layout, face_type(d), ncomp, 0); ed[d] = &edge[d]; }
    }
    GodParams P = make_params(g, dt, ncomp, bc, iconserv, is_velocity, use_forces_in_trans, force != nullptr, divu != nullptr);
    for (int d = 0; d < 3; ++d) {
        Tiling t = face_tiling(l, d, 1, 4);
        hipLaunchKernelGGL((k_trace<false>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, d, S.d_tab,
                           force ? force->d_tab : nullptr, umac[d]->d_tab, e0[d].d_tab, P);
    }
    for (int d = 0; d < 3; ++d) {
        Tiling t = face_tiling(l, d, 0, 4);
        hipLaunchKernelGGL((k_final<false>), t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, d, S.d_tab,
                           force ? force->d_tab : nullptr, divu ? divu->d_tab : nullptr,
                           umac[0]->d_tab, umac[1]->d_tab, umac[2]->d_tab, e0[0].d_tab, e0[1].d_tab, e0[2].d_tab, ed[d]->d_tab, P);
    }
    Tiling t = level_tiling(l, cell_type(), 0, 4);
    const bool sf = flux_out && flux_out[0];
    hipLaunchKernelGGL(k_aofs, t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, aofs.d_tab, acomp,
                       ed[0]->d_tab, ed[1]->d_tab, ed[2]->d_tab, umac[0]->d_tab, umac[1]->d_tab, umac[2]->d_tab,
                       sf ? flux_out[0]->d_tab : nullptr, sf ? flux_out[1]->d_tab : nullptr, sf ? flux_out[2]->d_tab : nullptr, P);
}

}  // namespace iamrx
