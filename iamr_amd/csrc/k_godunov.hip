// iamr_amd/csrc/k_godunov.hip -- Godunov (PLM + corner-transport-upwind) kernels for gfx950.
//
// Role: AMReX-Hydro Godunov::ExtrapVelToFaces (reference call site Source/NavierStokesBase.cpp:4487-4491)
// and HydroUtils::ComputeFluxesOnBoxFromState / ComputeDivergence / ComputeConvectiveTerm
// (Source/NavierStokesBase.cpp:4701-4842), SURVEY a3 / a8.
//
// MI355X-first structure (NOT the reference's ~20 scratch arrays per box).  PLM, the default: ONE fused z-marching launch per
// operation -- k_god_z (advection: edge states, fluxes, aofs) and k_pred_z (ExtrapVelToFaces) further down: the z-stencil of a
// column in registers, every intermediate of the corner-transport scheme in LDS ring planes, nothing but inputs and results in HBM.
// PPM (and IAMRX_GODUNOV_Z=0, the reference of tests/test_gpu_godunov_fused.py): the multi-pass kernels
//   pass 1  k_trace   : per face of each direction (transverse grown by 1): 4th-order limited slopes are
//                       evaluated in registers, the two traced states are upwinded at once ->
//                       advective velocity (predict mode) + single-valued transverse states.
//   pass 2  k_final   : per face: the four corner-coupled transverse states per transverse direction are
//                       rebuilt in registers from pass-1 data (re-evaluating slopes instead of storing
//                       lo/hi arrays), transverse terms + forcing + BCs + final upwinding.
//   pass 3  k_aofs    : per cell: area-weighted fluxes, -div, convective correction, aofs = -update.
// Only 3+3*ncomp face arrays are materialised in HBM between the passes.
// The direction is a template parameter and all index shifts are linear strides, so every array index is
// static after unrolling (no scratch memory); run-time parameters live in a small device-resident block.
#include "kernels.h"
#include "launch.h"
#include <cstdlib>
#include <cstddef>
#include <type_traits>

namespace iamrx {

#define SMALL_VEL 1.e-8
// x / dx in the fused kernels: a true division in the bit-reproducible build (STRICT_FP=1, the oracle's arithmetic), a multiplication
// with the reciprocal otherwise (an fp64 division is ~12 vector instructions; the difference is one rounding, inside the 1e-13 bar)
#ifdef IAMRX_STRICT_FP
#define IAMRX_DIVDX(x, dx, rdx) ((x) / (dx))
#else
#define IAMRX_DIVDX(x, dx, rdx) ((x) * (rdx))
#endif

constexpr int GOD_MAXC = 8;    // components of one call: the whole state (3 velocities + density + tracer(s) + temperature) at most

struct GodBC {
    int dlo[3], dhi[3];
    int per[3];
    BCRec bc[GOD_MAXC];
};

struct GodParams {
    double dt;
    double dx[3];
    int ncomp;
    int is_velocity;
    int fit;            // use_forces_in_trans
    int has_force;
    int has_divu;
    int iconserv[GOD_MAXC];
    GodBC bc;
};

__device__ __forceinline__ double lim2(double dlft, double drgt)
{
    const double dcen = 0.5 * (dlft + drgt);
    const double dsgn = copysign(1.0, dcen);
    const double dlim = (dlft * drgt >= 0.0) ? 2.0 * fmin(fabs(dlft), fabs(drgt)) : 0.0;
    return dsgn * fmin(dlim, fabs(dcen));
}

// amrex_calc_{x,y,z}slope_extdir, order 4, on the five values q(i-2..i+2); i = cell index in the slope direction.
__device__ __forceinline__ double slope4v(double qmm, double qm, double qi, double qp, double qpp, bool edlo, bool edhi, int i, int domlo, int domhi)
{
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
// ---- limited slopes of a thread row (fused z-marching kernels, no boundary-condition variants).  slope4v(i) evaluates three second-order
// limited differences: lim2 at the cells i - 1 and i + 1 and the ingredients of lim2 at i.  In a row of 16 lanes that own consecutive cells
// every lane forms lim2 of ITS cell once and takes its neighbours' through DPP row shifts (v_mov_b32_dpp row_shr:1 / row_shl:1: vector moves,
// no LDS, no barrier); the first and the last column of the row evaluate the one lim2 outside the row themselves.  The slope of the cell to
// the left (the low-side state of the lane's face) is the left lane's slope.  Same operands and expressions as slope4v: the same doubles.
// (measured at 256^3, MI355X: ExtrapVelToFaces 1.02 -> 1.00 ms, ComputeAofs of the velocity 1.845 -> 1.877 ms -- the instructions saved are
// not what bounds the kernels (two wavefronts per SIMD, four barriers per plane): kept as a build option, off)
#ifndef IAMRX_GOD_ROW16
#define IAMRX_GOD_ROW16 0
#endif
__device__ __forceinline__ double row16_lo(double v)         // the value lane - 1 of the 16-lane row holds (lane 0: its own)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_hi(double v)         // ... lane + 1 (lane 15: its own)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x101, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x101, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// slh: slope4v(a2, a1, c0, b1, b2) of the lane's cell, sll: of the cell to its left (garbage in the row's first lane: the low-side state of
// a face nobody uses); first / last: the lane owns the first / last column of the row that is in use
__device__ __forceinline__ void slope4_row16(double a2, double a1, double c0, double b1, double b2, bool first, bool last, double& sll, double& slh)
{
    const double dlft = c0 - a1, drgt = b1 - c0;
    const double dcen = 0.5 * (dlft + drgt);
    const double dsgn = copysign(1.0, dcen);
    const double dlim = (dlft * drgt >= 0.0) ? 2.0 * fmin(fabs(dlft), fabs(drgt)) : 0.0;
    const double d2 = dsgn * fmin(dlim, fabs(dcen));                   // lim2(dlft, drgt)
    // the one limited difference outside the row: cell i - 1 for the first column, cell i + 1 for the last one
    const double e0 = first ? a2 : c0, e1 = first ? a1 : b1, e2 = first ? c0 : b2;
    const double de = lim2(e1 - e0, e2 - e1);
    const double dm = row16_lo(d2), dp = row16_hi(d2);
    const double dfm = first ? de : dm, dfp = last ? de : dp;
    const double dtemp = 4.0 / 3.0 * dcen - 1.0 / 6.0 * (dfp + dfm);
    slh = dsgn * fmin(dlim, fabs(dtemp));
    sll = row16_lo(slh);
}
// q points at the cell, s = stride in the slope direction
template <class P>
__device__ __forceinline__ double slope4(P q, long s, bool edlo, bool edhi, int i, int domlo, int domhi)
{
    return slope4v(q[-2 * s], q[-s], q[0], q[s], q[2 * s], edlo, edhi, i, domlo, domhi);
}

__device__ __forceinline__ bool ed_or_ho(int b) { return b == bc_ext_dir || b == bc_hoextrap; }

// ---- PPM (ns.advection_scheme = Godunov_PPM: reference Source/NavierStokesBase.cpp:548-553, 4654-4656 pass use_ppm to AMReX-Hydro's
// "Godunov").  Colella & Woodward's piecewise-parabolic reconstruction as hydro_godunov_ppm states it (upstream; restated from the
// published algorithm): van Leer slopes -> 4th-order edge values clipped to the two neighbouring cells -> one-sided edge values at
// ext_dir / hoextrap faces -> monotonisation -> parabola integrated over the domain of dependence |u| dt of the face.
// Selected per call through a device-resident switch (every traced state of every pass goes through trace_lohi).
__device__ int g_ppm_dev = 0;

__device__ __forceinline__ double vanleer(double s0, double sm1, double sp1)
{
    const double dsc = 0.5 * (sp1 - sm1), dsl = 2.0 * (s0 - sm1), dsr = 2.0 * (sp1 - s0);
    return (dsl * dsr > 0.0) ? copysign(1.0, dsc) * fmin(fabs(dsc), fmin(fabs(dsl), fabs(dsr))) : 0.0;
}
__device__ __forceinline__ double clip2(double v, double a, double b) { return fmax(fmin(v, fmax(a, b)), fmin(a, b)); }

// monotonised edge values (sm: low face, sp: high face) of the cell q points at; i = its index in the direction of stride s
template <class P>
__device__ __forceinline__ void ppm_edges(P q, long s, bool edlo, bool edhi, int i, int domlo, int domhi, double& sm, double& sp)
{
    const double s0 = q[0], sm1 = q[-s], sm2 = q[-2 * s], sp1 = q[s], sp2 = q[2 * s];
    const double dm = vanleer(sm1, sm2, s0), d0 = vanleer(s0, sm1, sp1), dp = vanleer(sp1, s0, sp2);
    sm = clip2(0.5 * (s0 + sm1) - (1.0 / 6.0) * (d0 - dm), s0, sm1);
    sp = clip2(0.5 * (sp1 + s0) - (1.0 / 6.0) * (dp - d0), sp1, s0);
    if (edlo && (i == domlo || i == domlo + 1)) {
        const long o = (long)(domlo - i) * s;       // offset of cell domlo
        const double sb = q[o - s], a0 = q[o], a1 = q[o + s], a2 = q[o + 2 * s];
        const double se = clip2(-0.2 * sb + 0.75 * a0 + 0.5 * a1 - 0.05 * a2, a1, a0);
        if (i == domlo) { sm = sb; sp = se; } else sm = se;
    }
    if (edhi && (i == domhi || i == domhi - 1)) {
        const long o = (long)(domhi - i) * s;
        const double sb = q[o + s], a0 = q[o], a1 = q[o - s], a2 = q[o - 2 * s];
        const double se = clip2(-0.2 * sb + 0.75 * a0 + 0.5 * a1 - 0.05 * a2, a1, a0);
        if (i == domhi) { sp = sb; sm = se; } else sp = se;
    }
    if ((sp - s0) * (s0 - sm) <= 0.0) { sp = s0; sm = s0; }
    else if (fabs(sp - s0) >= 2.0 * fabs(sm - s0)) sp = 3.0 * s0 - 2.0 * sm;
    else if (fabs(sm - s0) >= 2.0 * fabs(sp - s0)) sm = 3.0 * s0 - 2.0 * sp;
}
// state on the high (side) / low face of a cell with value s0 and monotonised edge values sm, sp, traced with velocity u
__device__ __forceinline__ double ppm_state(double s0, double sm, double sp, double u, double dtdx, bool side)
{
    const double s6 = 6.0 * s0 - 3.0 * (sm + sp), sigma = fabs(u) * dtdx;
    if (side) return (u > SMALL_VEL) ? sp - (0.5 * sigma) * ((sp - sm) - (1.0 - (2.0 / 3.0) * sigma) * s6) : s0;
    return (u < -SMALL_VEL) ? sm + (0.5 * sigma) * ((sp - sm) + (1.0 - (2.0 / 3.0) * sigma) * s6) : s0;
}
template <class P>
__device__ __forceinline__ double ppm_trace(P q, long s, bool edlo, bool edhi, int i, int domlo, int domhi, double u, double dtdx, bool side)
{
    double sm, sp;
    ppm_edges(q, s, edlo, edhi, i, domlo, domhi, sm, sp);
    return ppm_state(q[0], sm, sp, u, dtdx, side);
}

// SetTransTerm{X,Y,Z}BCs.  qc points at the state value of the cell on the HIGH side of the face
// (cell index f in direction d), s = stride in d, f = face index.
template <class P>
__device__ __forceinline__ void trans_bc(P qc, long s, int f, bool normal_vel, double& lo, double& hi,
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
template <class P>
__device__ __forceinline__ void edge_bc(P qc, long s, int f, bool normal_vel, double& lo, double& hi,
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

// value forms of trans_bc / edge_bc for a face f with domlo <= f <= domhi + 1: the ext_dir data of the low boundary sits in the
// low-side cell of the face (qlc), that of the high boundary in the high-side cell (qhc)
__device__ __forceinline__ void trans_bc_v(double qlc, double qhc, int f, bool normal_vel, double& lo, double& hi,
                                           int bclo, int bchi, int domlo, int domhi)
{
    if (f <= domlo) {
        if (bclo == bc_ext_dir) { lo = qlc; if (normal_vel) hi = lo; }
        else if (bclo == bc_foextrap || bclo == bc_hoextrap || bclo == bc_reflect_even) lo = hi;
        else if (bclo == bc_reflect_odd) { hi = 0.; lo = 0.; }
    } else if (f > domhi) {
        if (bchi == bc_ext_dir) { hi = qhc; if (normal_vel) lo = hi; }
        else if (bchi == bc_foextrap || bchi == bc_hoextrap || bchi == bc_reflect_even) hi = lo;
        else if (bchi == bc_reflect_odd) { lo = 0.; hi = 0.; }
    }
}
__device__ __forceinline__ void edge_bc_v(double qlc, double qhc, int f, bool normal_vel, double& lo, double& hi,
                                          int bclo, int bchi, int domlo, int domhi)
{
    if (f <= domlo) {
        if (bclo == bc_ext_dir) { lo = qlc; if (normal_vel) hi = lo; }
        else if (bclo == bc_foextrap || bclo == bc_hoextrap || bclo == bc_reflect_even) {
            if (normal_vel && bclo != bc_reflect_even) hi = fmin(hi, 0.);
            lo = hi;
        } else if (bclo == bc_reflect_odd) { hi = 0.; lo = 0.; }
    } else if (f > domhi) {
        if (bchi == bc_ext_dir) { hi = qhc; if (normal_vel) lo = hi; }
        else if (bchi == bc_foextrap || bchi == bc_hoextrap || bchi == bc_reflect_even) {
            if (normal_vel && bchi != bc_reflect_even) lo = fmax(lo, 0.);
            hi = lo;
        } else if (bchi == bc_reflect_odd) { lo = 0.; hi = 0.; }
    }
}

// single-valued state of a face from its two traced states and the advecting velocity
__device__ __forceinline__ double upwind_fu(double um, double l, double h)
{
    const double st = (um >= 0.) ? l : h;
    const double fu = (fabs(um) < SMALL_VEL) ? 0.0 : 1.0;
    return fu * st + (1.0 - fu) * 0.5 * (h + l);
}

// second half of corner_state on values: transverse correction of the traced states l, h of a T-face by the O-direction, BCs, upwinding.
// mo_* / eo_*: mac velocity / pass-1 state on the low (cm, f) and high (cmo, fo) O-face of the low-side / high-side cell of the T-face.
__device__ __forceinline__ double corner_core(double l, double h, double qlc, double qhc, double macT_f,
    double mo_cm, double mo_cmo, double mo_f, double mo_fo, double eo_cm, double eo_cmo, double eo_f, double eo_fo,
    double dvl, double dvh, bool conserv, double c_o, double dt3, double dxO,
    bool nonperT, bool normal_vel, int fT, int bl, int bh, int domlo, int domhi)
{
    if (conserv) {
        const double rdxO = 1.0 / dxO;      // uniform: hoisted out of the plane loop
        l = l - c_o * (eo_cmo * mo_cmo - eo_cm * mo_cm) + dt3 * qlc * (IAMRX_DIVDX(mo_cmo - mo_cm, dxO, rdxO) - 0.5 * dvl);
        h = h - c_o * (eo_fo * mo_fo - eo_f * mo_f) + dt3 * qhc * (IAMRX_DIVDX(mo_fo - mo_f, dxO, rdxO) - 0.5 * dvh);
    } else {
        l = l - c_o * (mo_cmo + mo_cm) * (eo_cmo - eo_cm);
        h = h - c_o * (mo_fo + mo_f) * (eo_fo - eo_f);
    }
    if (nonperT) trans_bc_v(qlc, qhc, fT, normal_vel, l, h, bl, bh, domlo, domhi);
    return upwind_fu(macT_f, l, h);
}

// second half of final_edge on values (advection form): transverse terms from the corner-coupled states A?? / B?? (l / h: low-side / high-side
// cell of the D-face, 0 / 1: its low / high TA- or TB-face), forcing, BCs, upwinding with the mac velocity umD (PRED: with itself)
template <bool PRED>
__device__ __forceinline__ double final_core(double stl, double sth, double umD,
    double mA_l0, double mA_l1, double mA_h0, double mA_h1, double mB_l0, double mB_l1, double mB_h0, double mB_h1,
    double Al0, double Al1, double Ah0, double Ah1, double Bl0, double Bl1, double Bh0, double Bh1,
    double qlc, double qhc, double frl, double frh, double dvl, double dvh,
    bool conserv, bool has_divu, bool late_force, double dt, double dxA, double dxB,
    bool nonperD, bool normal_vel, int f, int blD, int bhD, int dloD, int dhiD)
{
    const double hdt = 0.5 * dt;
    if (conserv) {
        const double cfA = 0.5 * dt / dxA, cfB = 0.5 * dt / dxB;
        stl += -cfA * (Al1 * mA_l1 - Al0 * mA_l0);
        sth += -cfA * (Ah1 * mA_h1 - Ah0 * mA_h0);
        stl += -cfB * (Bl1 * mB_l1 - Bl0 * mB_l0);
        sth += -cfB * (Bh1 * mB_h1 - Bh0 * mB_h0);
        stl += cfA * qlc * (mA_l1 - mA_l0);
        sth += cfA * qhc * (mA_h1 - mA_h0);
        stl += cfB * qlc * (mB_l1 - mB_l0);
        sth += cfB * qhc * (mB_h1 - mB_h0);
        if (has_divu) { stl -= 0.5 * dt * qlc * dvl; sth -= 0.5 * dt * qhc * dvh; }
    } else {
        const double cfA = 0.25 * dt / dxA, cfB = 0.25 * dt / dxB;
        stl -= cfA * (mA_l1 + mA_l0) * (Al1 - Al0);
        sth -= cfA * (mA_h1 + mA_h0) * (Ah1 - Ah0);
        stl -= cfB * (mB_l1 + mB_l0) * (Bl1 - Bl0);
        sth -= cfB * (mB_h1 + mB_h0) * (Bh1 - Bh0);
    }
    if (late_force) { stl += hdt * frl; sth += hdt * frh; }
    if (nonperD) edge_bc_v(qlc, qhc, f, normal_vel, stl, sth, blD, bhD, dloD, dhiD);
    if (PRED) {
        const double st = ((stl + sth) >= 0.) ? stl : sth;
        const bool ltm = ((stl <= 0. && sth >= 0.) || (fabs(stl + sth) < SMALL_VEL));
        return ltm ? 0. : st;
    }
    double temp = (umD >= 0.) ? stl : sth;
    temp = (fabs(umD) < SMALL_VEL) ? 0.5 * (stl + sth) : temp;
    return temp;
}

// traced states from the limited slopes slh (cell f) and sll (cell f-1)
template <bool PRED, class P, class PV>
__device__ __forceinline__ void trace_from_slopes(P qn, PV vd, long s, double um, double dtdx, double slh, double sll, double& lo, double& hi)
{
    if constexpr (PRED) {
        hi = qn[0] + 0.5 * (-1.0 - vd[0] * dtdx) * slh;
        lo = qn[-s] + 0.5 * (1.0 - vd[-s] * dtdx) * sll;
    } else {
        hi = qn[0] + 0.5 * (-1.0 - um * dtdx) * slh;
        lo = qn[-s] + 0.5 * (1.0 - um * dtdx) * sll;
    }
}

// traced states on face f of direction d for component n: lo from cell f-1, hi from cell f.
// PRED: trace velocity = cell-centred vcc(cell, d); else the face's own mac velocity `um`.
template <bool PRED, class P, class PV>
__device__ __forceinline__ void trace_lohi(P qn /*state comp n at cell f*/, PV vd /*vcc comp d at cell f (PRED)*/,
                                           long s, double um, double dtdx, bool edlo, bool edhi, int f, int domlo, int domhi,
                                           double& lo, double& hi)
{
    if (g_ppm_dev) {
        // lo: high face of cell f-1, hi: low face of cell f; PRED traces with the cell-centred velocity of each cell
        double ul = um, uh = um;
        if constexpr (PRED) { ul = vd[-s]; uh = vd[0]; }
        lo = ppm_trace(qn - s, s, edlo, edhi, f - 1, domlo, domhi, ul, dtdx, true);
        hi = ppm_trace(qn, s, edlo, edhi, f, domlo, domhi, uh, dtdx, false);
        return;
    }
    const double slh = slope4(qn, s, edlo, edhi, f, domlo, domhi);
    const double sll = slope4(qn - s, s, edlo, edhi, f - 1, domlo, domhi);
    trace_from_slopes<PRED>(qn, vd, s, um, dtdx, slh, sll, lo, hi);
}

// traced states with the slopes taken from the slope array written by k_trace (slp at cell f, stride ss); slp == nullptr: compute
template <bool PRED, class P, class PV, class PS>
__device__ __forceinline__ void trace_lohi_sl(P qn, PV vd, long s, double um, double dtdx, bool edlo, bool edhi, int f, int domlo, int domhi,
                                              PS slp, long ss, double& lo, double& hi)
{
    if constexpr (std::is_same<PS, std::nullptr_t>::value) trace_lohi<PRED>(qn, vd, s, um, dtdx, edlo, edhi, f, domlo, domhi, lo, hi);
    else {
        if (g_ppm_dev) trace_lohi<PRED>(qn, vd, s, um, dtdx, edlo, edhi, f, domlo, domhi, lo, hi);      // no slope arrays with PPM
        else trace_from_slopes<PRED>(qn, vd, s, um, dtdx, slp[0], slp[-ss], lo, hi);
    }
}

template <int D> __device__ __forceinline__ long stride_of(const FabD& a)
{
    return D == 0 ? 1L : (D == 1 ? (long)a.n[0] : (long)a.n[0] * a.n[1]);
}

// -------------------------------------------------------------------------------- pass 1
// grid over faces of direction D, transverse directions grown by 1.
// PRED: writes ad[D] (1 comp) and e0[D] (ncomp comps, upwinded with ad).  ADV: mac given, writes e0[D].
template <bool PRED, int D>
__global__ void __launch_bounds__(256) k_trace(Tiling t, const BoxD* __restrict__ boxes,
    const FabD* __restrict__ qt, const FabD* __restrict__ ft, const FabD* __restrict__ mact /*ADV: umac[D]; PRED: out ad[D]*/,
    const FabD* __restrict__ e0t, const FabD* __restrict__ slt /*optional out: limited slopes in D, cells grown by 1*/,
    const GodParams* __restrict__ Pp)
{
    const GodParams& P = *Pp;
    const int fab = blockIdx.y;
    BoxD b = boxes[fab];
#pragma unroll
    for (int e = 0; e < 3; ++e) { if (e == D) b.hi[e] += 1; else { b.lo[e] -= 1; b.hi[e] += 1; } }
    const int flo = b.lo[D];
    int i, j, k0, k1;
    if (!tile_ijk(t, b, i, j, k0, k1)) return;
    const FabD q = qt[fab], e0 = e0t[fab], mac = mact[fab];
    const bool has_force = P.has_force != 0, fit = P.fit != 0;
    FabD frc; if (has_force) frc = ft[fab];
    FabD sl; if (slt) sl = slt[fab];
    const long s = stride_of<D>(q);
    const long fs = has_force ? stride_of<D>(frc) : 0;
    const double hdt = 0.5 * P.dt;
    const double dtdx = P.dt / P.dx[D];
    const int domlo = P.bc.dlo[D], domhi = P.bc.dhi[D];
    const bool nonper = !P.bc.per[D];
    const int ncomp = P.ncomp;
    const bool is_vel = P.is_velocity != 0;
    for (int k = k0; k <= k1; ++k) {
        const int f = D == 0 ? i : (D == 1 ? j : k);
        const long qo = q.off(i, j, k);
        const long fo = has_force ? frc.off(i, j, k) : 0;
        double uad;
        if (PRED) {
            // advective velocity from the traced normal component
            const int bl = P.bc.bc[D].lo[D], bh = P.bc.bc[D].hi[D];
            const bool edlo = nonper && ed_or_ho(bl), edhi = nonper && ed_or_ho(bh);
            double l, h;
            trace_lohi<true>(q.gp() + qo + q.cs * D, q.gp() + qo + q.cs * D, s, 0.0, dtdx, edlo, edhi, f, domlo, domhi, l, h);
            if (fit && has_force) { l += hdt * frc.gp()[fo - fs + frc.cs * D]; h += hdt * frc.gp()[fo + frc.cs * D]; }
            if (nonper) trans_bc(q.gp() + qo + q.cs * D, s, f, is_vel, l, h, bl, bh, domlo, domhi);
            const double st = ((l + h) >= 0.) ? l : h;
            const bool ltm = ((l <= 0. && h >= 0.) || (fabs(l + h) < SMALL_VEL));
            uad = ltm ? 0. : st;
            mac(i, j, k, 0) = uad;
        } else uad = mac(i, j, k, 0);
        const double fu = (fabs(uad) < SMALL_VEL) ? 0.0 : 1.0;
        for (int n = 0; n < ncomp; ++n) {
            const int bl = P.bc.bc[n].lo[D], bh = P.bc.bc[n].hi[D];
            const bool edlo = nonper && ed_or_ho(bl), edhi = nonper && ed_or_ho(bh);
            double l, h;
            {
                const auto qn = q.gp() + qo + q.cs * n;
                if (g_ppm_dev) trace_lohi<PRED>(qn, q.gp() + qo + q.cs * D, s, uad, dtdx, edlo, edhi, f, domlo, domhi, l, h);
                else {
                const double slh = slope4(qn, s, edlo, edhi, f, domlo, domhi);
                const double sll = slope4(qn - s, s, edlo, edhi, f - 1, domlo, domhi);
                trace_from_slopes<PRED>(qn, q.gp() + qo + q.cs * D, s, uad, dtdx, slh, sll, l, h);
                if (slt) {
                    // by-product: the limited slope of cell f (and of cell f-1 from the first face) for the later passes
                    const long so = sl.off(i, j, k) + sl.cs * n;
                    sl.gp()[so] = slh;
                    if (f == flo) sl.gp()[so - stride_of<D>(sl)] = sll;
                }
                }
            }
            if (fit && has_force) { l += hdt * frc.gp()[fo - fs + frc.cs * n]; h += hdt * frc.gp()[fo + frc.cs * n]; }
            if (nonper) trans_bc(q.gp() + qo + q.cs * n, s, f, is_vel && n == D, l, h, bl, bh, domlo, domhi);
            const double st = (uad >= 0.) ? l : h;
            e0(i, j, k, n) = fu * st + (1.0 - fu) * 0.5 * (h + l);
        }
    }
}

// corner-coupled, upwinded state on one T-face.  All pointers are already positioned:
//   qn   : state comp n at the cell on the high side of the T-face (cell index == face index fT)
//   vT   : vcc comp T at the same cell (PRED only)
//   macO : mac[O] at that cell's low O-face;  eO : pass-1 state on that O-face (comp n)
template <bool PRED, class PQ, class PV, class PM, class PE, class PF, class PD, class PS = std::nullptr_t>
__device__ __forceinline__ double corner_state(PQ qn, PV vT, long sT, int fT,
    double macT_f, PM macO, long mOsT, long mOsO,
    PE eO, long eOsT, long eOsO,
    PF frcn, long fsT, double dtdxT, double c_o /* dt/(6 dxO) or dt/(3 dxO) */, double dt3, double dxO,
    bool conserv, PD divu, long dsT,
    bool fit, double hdt, bool nonperT, bool normal_vel, int bl, int bh, int domlo, int domhi, PS slp = nullptr, long ssT = 0)
{
    const bool edlo = nonperT && ed_or_ho(bl), edhi = nonperT && ed_or_ho(bh);
    double l, h;
    trace_lohi_sl<PRED>(qn, vT, sT, macT_f, dtdxT, edlo, edhi, fT, domlo, domhi, slp, ssT, l, h);
    if (fit && frcn) { l += hdt * frcn[-fsT]; h += hdt * frcn[0]; }
    if (nonperT) trans_bc(qn, sT, fT, normal_vel, l, h, bl, bh, domlo, domhi);   // BCs of the traced states (pass-1 order)
    const double mo_cm = macO[-mOsT], mo_cmo = macO[-mOsT + mOsO], mo_f = macO[0], mo_fo = macO[mOsO];
    const double eo_cm = eO[-eOsT], eo_cmo = eO[-eOsT + eOsO], eo_f = eO[0], eo_fo = eO[eOsO];
    if (conserv) {
        const double dvl = divu ? divu[-dsT] : 0.0, dvh = divu ? divu[0] : 0.0;
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

// the four corner states (low-side cell / high-side cell of the D-face) x (T-face c / c+1) for transverse direction T
template <bool PRED, int D, int T, class PQ, class PF, class PD>
__device__ __forceinline__ void corner_quad(const GodParams& P, int n, bool conserv, int fT,
    PQ qn, PQ qT, const FabD& q,
    const FabD& mT, long mTo, const FabD& mO, long mOo, const FabD& eO, long eOo,
    PF frcn, const FabD& frc, PD dvp, const FabD& dv,
    double& Tl0, double& Tl1, double& Th0, double& Th1)
{
    constexpr int O = 3 - D - T;
    const long qsD = stride_of<D>(q), qsT = stride_of<T>(q);
    const long mTsD = stride_of<D>(mT), mTsT = stride_of<T>(mT);
    const long mOsD = stride_of<D>(mO), mOsT = stride_of<T>(mO), mOsO = stride_of<O>(mO);
    const long eOsD = stride_of<D>(eO), eOsT = stride_of<T>(eO), eOsO = stride_of<O>(eO);
    const long fsD = frcn ? stride_of<D>(frc) : 0, fsT = frcn ? stride_of<T>(frc) : 0;
    const long dsD = dvp ? stride_of<D>(dv) : 0, dsT = dvp ? stride_of<T>(dv) : 0;
    const double c_o = conserv ? P.dt / (3.0 * P.dx[O]) : P.dt / (6.0 * P.dx[O]);
    const double dt3 = P.dt / 3.0;
    const double dtdxT = P.dt / P.dx[T];
    const double hdt = 0.5 * P.dt;
    const int bl = P.bc.bc[n].lo[T], bh = P.bc.bc[n].hi[T];
    const bool nonperT = !P.bc.per[T];
    const bool nvel = P.is_velocity && n == T;
    const bool fit = P.fit != 0;
    const int dlo = P.bc.dlo[T], dhi = P.bc.dhi[T];
#define IAMRX_CORNER(SIDE, UP)                                                                                                   \
    corner_state<PRED>(qn + ((SIDE) ? 0 : -qsD) + (UP) * qsT, qT + ((SIDE) ? 0 : -qsD) + (UP) * qsT, qsT, fT + (UP),                \
                       mT.gp()[mTo + ((SIDE) ? 0 : -mTsD) + (UP) * mTsT], mO.gp() + mOo + ((SIDE) ? 0 : -mOsD) + (UP) * mOsT, mOsT, mOsO, \
                       eO.gp() + eOo + ((SIDE) ? 0 : -eOsD) + (UP) * eOsT, eOsT, eOsO,                                                \
                       frcn ? frcn + ((SIDE) ? 0 : -fsD) + (UP) * fsT : nullptr, fsT, dtdxT, c_o, dt3, P.dx[O], conserv,             \
                       dvp ? dvp + ((SIDE) ? 0 : -dsD) + (UP) * dsT : nullptr, dsT, fit, hdt, nonperT, nvel, bl, bh, dlo, dhi)
    Tl0 = IAMRX_CORNER(0, 0);
    Tl1 = IAMRX_CORNER(0, 1);
    Th0 = IAMRX_CORNER(1, 0);
    Th1 = IAMRX_CORNER(1, 1);
#undef IAMRX_CORNER
}

// -------------------------------------------------------------------------------- pass 2
// grid over the valid faces of direction D.  PRED: only component n = D, output umac[D];
// ADV: all components, output final edge states edge[D](ncomp).
template <bool PRED, int D>
__global__ void __launch_bounds__(256) k_final(Tiling t, const BoxD* __restrict__ boxes,
    const FabD* __restrict__ qt, const FabD* __restrict__ ft, const FabD* __restrict__ divut,
    const FabD* __restrict__ m0t, const FabD* __restrict__ m1t, const FabD* __restrict__ m2t,
    const FabD* __restrict__ e0t, const FabD* __restrict__ e1t, const FabD* __restrict__ e2t,
    const FabD* __restrict__ outt, const GodParams* __restrict__ Pp)
{
    constexpr int TA = D == 0 ? 1 : 0;            // transverse directions in ascending order
    constexpr int TB = D == 2 ? 1 : 2;
    const GodParams& P = *Pp;
    const int fab = blockIdx.y;
    BoxD b = boxes[fab];
    b.hi[D] += 1;
    int i, j, k0, k1;
    if (!tile_ijk(t, b, i, j, k0, k1)) return;
    const FabD q = qt[fab], out = outt[fab];
    const FabD m0 = m0t[fab], m1 = m1t[fab], m2 = m2t[fab];
    const FabD e0 = e0t[fab], e1 = e1t[fab], e2 = e2t[fab];
    const FabD& mD = D == 0 ? m0 : (D == 1 ? m1 : m2);
    const FabD& mA = TA == 0 ? m0 : (TA == 1 ? m1 : m2);
    const FabD& mB = TB == 0 ? m0 : (TB == 1 ? m1 : m2);
    const FabD& eA = TA == 0 ? e0 : (TA == 1 ? e1 : e2);
    const FabD& eB = TB == 0 ? e0 : (TB == 1 ? e1 : e2);
    const bool has_force = P.has_force != 0, has_divu = P.has_divu != 0, fit = P.fit != 0;
    FabD frc; if (has_force) frc = ft[fab];
    FabD dv; if (has_divu) dv = divut[fab];
    const long qsD = stride_of<D>(q);
    const long fsD = has_force ? stride_of<D>(frc) : 0, dsD = has_divu ? stride_of<D>(dv) : 0;
    const long mAsD = stride_of<D>(mA), mAsT = stride_of<TA>(mA), mBsD = stride_of<D>(mB), mBsT = stride_of<TB>(mB);
    const double dt = P.dt, hdt = 0.5 * P.dt;
    const double dtdxD = dt / P.dx[D];
    const int nbeg = PRED ? D : (gridDim.z > 1 ? (int)blockIdx.z : 0), nend = PRED ? D + 1 : (gridDim.z > 1 ? nbeg + 1 : P.ncomp);
    const bool nonperD = !P.bc.per[D];
    const int dloD = P.bc.dlo[D], dhiD = P.bc.dhi[D];
    const bool is_vel = P.is_velocity != 0;
    for (int k = k0; k <= k1; ++k) {
        const int f = D == 0 ? i : (D == 1 ? j : k);
        const int fA = TA == 0 ? i : (TA == 1 ? j : k), fB = TB == 0 ? i : (TB == 1 ? j : k);
        const long qo = q.off(i, j, k);
        const long fo = has_force ? frc.off(i, j, k) : 0;
        const long dvo = has_divu ? dv.off(i, j, k) : 0;
        const long mAo = mA.off(i, j, k), mBo = mB.off(i, j, k);
        const double umD = mD(i, j, k, 0);
        for (int n = nbeg; n < nend; ++n) {
            const auto qn = q.gp() + qo + q.cs * n;
            const auto frcn = has_force ? frc.gp() + fo + frc.cs * n : (decltype(frc.gp()))nullptr;
            const auto dvp = has_divu ? dv.gp() + dvo : (decltype(dv.gp()))nullptr;
            const bool conserv = !PRED && P.iconserv[n] != 0;
            const int blD = P.bc.bc[n].lo[D], bhD = P.bc.bc[n].hi[D];
            // own traced states along D
            double stl, sth;
            {
                const bool edlo = nonperD && ed_or_ho(blD), edhi = nonperD && ed_or_ho(bhD);
                trace_lohi<PRED>(qn, q.gp() + qo + q.cs * D, qsD, umD, dtdxD, edlo, edhi, f, dloD, dhiD, stl, sth);
                if (fit && has_force) { stl += hdt * frcn[-fsD]; sth += hdt * frcn[0]; }
                if (nonperD) trans_bc(qn, qsD, f, is_vel && n == D, stl, sth, blD, bhD, dloD, dhiD);
            }
            // corner-coupled transverse states: direction TA is corrected with the TB-derivative and vice versa
            double Al0, Al1, Ah0, Ah1, Bl0, Bl1, Bh0, Bh1;
            corner_quad<PRED, D, TA>(P, n, conserv, fA, qn, q.gp() + qo + q.cs * TA, q, mA, mAo, mB, mBo, eB, eB.off(i, j, k) + eB.cs * n,
                                     frcn, frc, dvp, dv, Al0, Al1, Ah0, Ah1);
            corner_quad<PRED, D, TB>(P, n, conserv, fB, qn, q.gp() + qo + q.cs * TB, q, mB, mBo, mA, mAo, eA, eA.off(i, j, k) + eA.cs * n,
                                     frcn, frc, dvp, dv, Bl0, Bl1, Bh0, Bh1);
            const double mA_l0 = mA.gp()[mAo - mAsD], mA_l1 = mA.gp()[mAo - mAsD + mAsT], mA_h0 = mA.gp()[mAo], mA_h1 = mA.gp()[mAo + mAsT];
            const double mB_l0 = mB.gp()[mBo - mBsD], mB_l1 = mB.gp()[mBo - mBsD + mBsT], mB_h0 = mB.gp()[mBo], mB_h1 = mB.gp()[mBo + mBsT];
            if (conserv) {
                const double cA = 0.5 * dt / P.dx[TA], cB = 0.5 * dt / P.dx[TB];
                stl += -cA * (Al1 * mA_l1 - Al0 * mA_l0);
                sth += -cA * (Ah1 * mA_h1 - Ah0 * mA_h0);
                stl += -cB * (Bl1 * mB_l1 - Bl0 * mB_l0);
                sth += -cB * (Bh1 * mB_h1 - Bh0 * mB_h0);
                stl += cA * qn[-qsD] * (mA_l1 - mA_l0);
                sth += cA * qn[0] * (mA_h1 - mA_h0);
                stl += cB * qn[-qsD] * (mB_l1 - mB_l0);
                sth += cB * qn[0] * (mB_h1 - mB_h0);
                if (has_divu) { stl -= 0.5 * dt * qn[-qsD] * dvp[-dsD]; sth -= 0.5 * dt * qn[0] * dvp[0]; }
            } else {
                const double cA = 0.25 * dt / P.dx[TA], cB = 0.25 * dt / P.dx[TB];
                stl -= cA * (mA_l1 + mA_l0) * (Al1 - Al0);
                sth -= cA * (mA_h1 + mA_h0) * (Ah1 - Ah0);
                stl -= cB * (mB_l1 + mB_l0) * (Bl1 - Bl0);
                sth -= cB * (mB_h1 + mB_h0) * (Bh1 - Bh0);
            }
            if (!fit && has_force) { stl += hdt * frcn[-fsD]; sth += hdt * frcn[0]; }
            if (nonperD) edge_bc(qn, qsD, f, is_vel && n == D, stl, sth, blD, bhD, dloD, dhiD);
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

// -------------------------------------------------------------------------------- pass 2a / 2b (split variant)
// 2a: k_corner<D,T> materialises the corner-coupled transverse states on the T-faces (cells grown by 1 in D)
// 2b: k_final_s<D> combines them.  Less register pressure and no slope recomputation per corner than the
// fully fused k_final above (which stays selectable with IAMRX_GODUNOV_FUSED=1 for A/B measurements).
// corner-coupled state of component n on the T-face (i,j,k), corrected with the O-derivative (O = the third direction)
template <bool PRED, int D, int T, bool SL = false>
__device__ __forceinline__ double corner_at(const GodParams& P, int n, int i, int j, int k, const FabD& q, const FabD& frc, const FabD& dv,
                                            const FabD& mT, const FabD& mO, const FabD& eO, bool has_force, bool has_divu,
                                            const FabD* slT = nullptr /*SL: slopes in direction T*/)
{
    constexpr int O = 3 - D - T;
    const bool fit = P.fit != 0;
    const long qsT = stride_of<T>(q);
    const long mOsT = stride_of<T>(mO), mOsO = stride_of<O>(mO), eOsT = stride_of<T>(eO), eOsO = stride_of<O>(eO);
    const long fsT = has_force ? stride_of<T>(frc) : 0, dsT = has_divu ? stride_of<T>(dv) : 0;
    const double dt3 = P.dt / 3.0, dtdxT = P.dt / P.dx[T], hdt = 0.5 * P.dt;
    const bool nonperT = !P.bc.per[T];
    const int dlo = P.bc.dlo[T], dhi = P.bc.dhi[T];
    const int fT = T == 0 ? i : (T == 1 ? j : k);
    const long qo = q.off(i, j, k);
    const double macT = mT(i, j, k, 0);
    const long mOo = mO.off(i, j, k), eOo = eO.off(i, j, k);
    const bool conserv = !PRED && P.iconserv[n] != 0;
    const double c_o = conserv ? P.dt / (3.0 * P.dx[O]) : P.dt / (6.0 * P.dx[O]);
    if constexpr (SL)
        return corner_state<PRED>(q.gp() + qo + q.cs * n, q.gp() + qo + q.cs * T, qsT, fT, macT, mO.gp() + mOo, mOsT, mOsO,
            eO.gp() + eOo + eO.cs * n, eOsT, eOsO, has_force ? frc.gp() + frc.off(i, j, k) + frc.cs * n : nullptr, fsT, dtdxT, c_o, dt3,
            P.dx[O], conserv, has_divu ? dv.gp() + dv.off(i, j, k) : nullptr, dsT, fit, hdt, nonperT, P.is_velocity && n == T,
            P.bc.bc[n].lo[T], P.bc.bc[n].hi[T], dlo, dhi, slT->gp() + slT->off(i, j, k) + slT->cs * n, stride_of<T>(*slT));
    else
    return corner_state<PRED>(q.gp() + qo + q.cs * n, q.gp() + qo + q.cs * T, qsT, fT, macT, mO.gp() + mOo, mOsT, mOsO,
        eO.gp() + eOo + eO.cs * n, eOsT, eOsO, has_force ? frc.gp() + frc.off(i, j, k) + frc.cs * n : nullptr, fsT, dtdxT, c_o, dt3,
        P.dx[O], conserv, has_divu ? dv.gp() + dv.off(i, j, k) : nullptr, dsT, fit, hdt, nonperT, P.is_velocity && n == T,
        P.bc.bc[n].lo[T], P.bc.bc[n].hi[T], dlo, dhi);
}

template <bool PRED, int D, int T>
__global__ void __launch_bounds__(256) k_corner(Tiling t, const BoxD* __restrict__ boxes,
    const FabD* __restrict__ qt, const FabD* __restrict__ ft, const FabD* __restrict__ divut,
    const FabD* __restrict__ mTt, const FabD* __restrict__ mOt, const FabD* __restrict__ eOt,
    const FabD* __restrict__ outt, const GodParams* __restrict__ Pp)
{
    const GodParams& P = *Pp;
    const int fab = blockIdx.y;
    BoxD b = boxes[fab];
    b.hi[T] += 1; b.lo[D] -= 1; b.hi[D] += 1;
    int i, j, k0, k1;
    if (!tile_ijk(t, b, i, j, k0, k1)) return;
    const FabD q = qt[fab], mT = mTt[fab], mO = mOt[fab], eO = eOt[fab], out = outt[fab];
    const bool has_force = P.has_force != 0, has_divu = P.has_divu != 0;
    FabD frc; if (has_force) frc = ft[fab];
    FabD dv; if (has_divu) dv = divut[fab];
    // advection: one component per blockIdx.z (fewer live registers, more workgroups); prediction: the normal component only
    const int nbeg = PRED ? D : (gridDim.z > 1 ? (int)blockIdx.z : 0), nend = PRED ? D + 1 : (gridDim.z > 1 ? nbeg + 1 : P.ncomp);
    for (int k = k0; k <= k1; ++k)
        for (int n = nbeg; n < nend; ++n)
            out(i, j, k, PRED ? 0 : n) = corner_at<PRED, D, T>(P, n, i, j, k, q, frc, dv, mT, mO, eO, has_force, has_divu);
}

// final edge state of component n on the D-face (i,j,k) from the two corner-coupled transverse arrays.  cAp / cBp point at the
// entry (i,j,k) of the TA- / TB-face corner states of component n; strides: s?D towards the low-side cell of the D-face,
// s?T to the next TA- / TB-face (global arrays in k_final_s, LDS ring planes in k_dir).  mA / mB: the four transverse mac
// velocities (low-side cell: face 0/1, high-side cell: face 0/1).
template <bool PRED, int D, bool SL = false, class PC>
__device__ __forceinline__ double final_edge(const GodParams& P, int n, int i, int j, int k, const FabD& q, const FabD& frc, const FabD& dv,
    bool has_force, bool has_divu, double umD, double mA_l0, double mA_l1, double mA_h0, double mA_h1,
    double mB_l0, double mB_l1, double mB_h0, double mB_h1, PC cAp, long cAsD, long cAsT, PC cBp, long cBsD, long cBsT,
    const FabD* slD = nullptr /*SL: slopes in direction D*/)
{
    constexpr int TA = D == 0 ? 1 : 0;
    constexpr int TB = D == 2 ? 1 : 2;
    const bool fit = P.fit != 0, is_vel = P.is_velocity != 0;
    const long qsD = stride_of<D>(q);
    const long fsD = has_force ? stride_of<D>(frc) : 0, dsD = has_divu ? stride_of<D>(dv) : 0;
    const double dt = P.dt, hdt = 0.5 * P.dt;
    const double dtdxD = dt / P.dx[D];
    const bool nonperD = !P.bc.per[D];
    const int dloD = P.bc.dlo[D], dhiD = P.bc.dhi[D];
    const int f = D == 0 ? i : (D == 1 ? j : k);
    const long qo = q.off(i, j, k);
    const long fo = has_force ? frc.off(i, j, k) : 0;
    const long dvo = has_divu ? dv.off(i, j, k) : 0;
    const auto qn = q.gp() + qo + q.cs * n;
    const auto frcn = has_force ? frc.gp() + fo + frc.cs * n : (decltype(frc.gp()))nullptr;
    const bool conserv = !PRED && P.iconserv[n] != 0;
    const int blD = P.bc.bc[n].lo[D], bhD = P.bc.bc[n].hi[D];
    double stl, sth;
    {
        const bool edlo = nonperD && ed_or_ho(blD), edhi = nonperD && ed_or_ho(bhD);
        if constexpr (SL)
            trace_lohi_sl<PRED>(qn, q.gp() + qo + q.cs * D, qsD, umD, dtdxD, edlo, edhi, f, dloD, dhiD,
                                slD->gp() + slD->off(i, j, k) + slD->cs * n, stride_of<D>(*slD), stl, sth);
        else
        trace_lohi<PRED>(qn, q.gp() + qo + q.cs * D, qsD, umD, dtdxD, edlo, edhi, f, dloD, dhiD, stl, sth);
        if (fit && has_force) { stl += hdt * frcn[-fsD]; sth += hdt * frcn[0]; }
        if (nonperD) trans_bc(qn, qsD, f, is_vel && n == D, stl, sth, blD, bhD, dloD, dhiD);
    }
    const double Al0 = cAp[-cAsD], Al1 = cAp[-cAsD + cAsT], Ah0 = cAp[0], Ah1 = cAp[cAsT];
    const double Bl0 = cBp[-cBsD], Bl1 = cBp[-cBsD + cBsT], Bh0 = cBp[0], Bh1 = cBp[cBsT];
    if (conserv) {
        const double cfA = 0.5 * dt / P.dx[TA], cfB = 0.5 * dt / P.dx[TB];
        stl += -cfA * (Al1 * mA_l1 - Al0 * mA_l0);
        sth += -cfA * (Ah1 * mA_h1 - Ah0 * mA_h0);
        stl += -cfB * (Bl1 * mB_l1 - Bl0 * mB_l0);
        sth += -cfB * (Bh1 * mB_h1 - Bh0 * mB_h0);
        stl += cfA * qn[-qsD] * (mA_l1 - mA_l0);
        sth += cfA * qn[0] * (mA_h1 - mA_h0);
        stl += cfB * qn[-qsD] * (mB_l1 - mB_l0);
        sth += cfB * qn[0] * (mB_h1 - mB_h0);
        if (has_divu) { stl -= 0.5 * dt * qn[-qsD] * dv.gp()[dvo - dsD]; sth -= 0.5 * dt * qn[0] * dv.gp()[dvo]; }
    } else {
        const double cfA = 0.25 * dt / P.dx[TA], cfB = 0.25 * dt / P.dx[TB];
        stl -= cfA * (mA_l1 + mA_l0) * (Al1 - Al0);
        sth -= cfA * (mA_h1 + mA_h0) * (Ah1 - Ah0);
        stl -= cfB * (mB_l1 + mB_l0) * (Bl1 - Bl0);
        sth -= cfB * (mB_h1 + mB_h0) * (Bh1 - Bh0);
    }
    if (!fit && has_force) { stl += hdt * frcn[-fsD]; sth += hdt * frcn[0]; }
    if (nonperD) edge_bc(qn, qsD, f, is_vel && n == D, stl, sth, blD, bhD, dloD, dhiD);
    if (PRED) {
        const double st = ((stl + sth) >= 0.) ? stl : sth;
        const bool ltm = ((stl <= 0. && sth >= 0.) || (fabs(stl + sth) < SMALL_VEL));
        return ltm ? 0. : st;
    }
    double temp = (umD >= 0.) ? stl : sth;
    temp = (fabs(umD) < SMALL_VEL) ? 0.5 * (stl + sth) : temp;
    return temp;
}

template <bool PRED, int D>
__global__ void __launch_bounds__(256) k_final_s(Tiling t, const BoxD* __restrict__ boxes,
    const FabD* __restrict__ qt, const FabD* __restrict__ ft, const FabD* __restrict__ divut,
    const FabD* __restrict__ mDt, const FabD* __restrict__ mAt, const FabD* __restrict__ mBt,
    const FabD* __restrict__ cAt, const FabD* __restrict__ cBt, const FabD* __restrict__ outt, const GodParams* __restrict__ Pp)
{
    constexpr int TA = D == 0 ? 1 : 0;
    constexpr int TB = D == 2 ? 1 : 2;
    const GodParams& P = *Pp;
    const int fab = blockIdx.y;
    BoxD b = boxes[fab];
    b.hi[D] += 1;
    int i, j, k0, k1;
    if (!tile_ijk(t, b, i, j, k0, k1)) return;
    const FabD q = qt[fab], out = outt[fab], mD = mDt[fab], mA = mAt[fab], mB = mBt[fab], cA = cAt[fab], cB = cBt[fab];
    const bool has_force = P.has_force != 0, has_divu = P.has_divu != 0;
    FabD frc; if (has_force) frc = ft[fab];
    FabD dv; if (has_divu) dv = divut[fab];
    const long mAsD = stride_of<D>(mA), mAsT = stride_of<TA>(mA), mBsD = stride_of<D>(mB), mBsT = stride_of<TB>(mB);
    const long cAsD = stride_of<D>(cA), cAsT = stride_of<TA>(cA), cBsD = stride_of<D>(cB), cBsT = stride_of<TB>(cB);
    const int nbeg = PRED ? D : (gridDim.z > 1 ? (int)blockIdx.z : 0), nend = PRED ? D + 1 : (gridDim.z > 1 ? nbeg + 1 : P.ncomp);
    for (int k = k0; k <= k1; ++k) {
        const long mAo = mA.off(i, j, k), mBo = mB.off(i, j, k);
        const double umD = mD(i, j, k, 0);
        const double mA_l0 = mA.gp()[mAo - mAsD], mA_l1 = mA.gp()[mAo - mAsD + mAsT], mA_h0 = mA.gp()[mAo], mA_h1 = mA.gp()[mAo + mAsT];
        const double mB_l0 = mB.gp()[mBo - mBsD], mB_l1 = mB.gp()[mBo - mBsD + mBsT], mB_h0 = mB.gp()[mBo], mB_h1 = mB.gp()[mBo + mBsT];
        for (int n = nbeg; n < nend; ++n) {
            const int cn = PRED ? 0 : n;
            out(i, j, k, cn) = final_edge<PRED, D>(P, n, i, j, k, q, frc, dv, has_force, has_divu, umD, mA_l0, mA_l1, mA_h0, mA_h1,
                                                  mB_l0, mB_l1, mB_h0, mB_h1, cA.gp() + cA.off(i, j, k) + cA.cs * cn, cAsD, cAsT,
                                                  cB.gp() + cB.off(i, j, k) + cB.cs * cn, cBsD, cBsT);
        }
    }
}

// -------------------------------------------------------------------------------- pass 2 fused per direction (default)
// k_dir<D>: the two corner-coupled arrays of direction D never reach HBM.  A workgroup owns a TX x TY tile of D-faces and
// marches through the planes k0..k1.  Its 14 wavefronts are specialised:
//   (TX x TY = 32 x 8: 14 wavefronts)
//   waves 0-4  (producer A): corner states on the TA-faces the tile needs (tile grown by one cell / one face), one per thread
//   waves 5-9  (producer B): the same for the TB-faces
//   waves 10-13 (consumer) : final edge state of one D-face per thread from the LDS ring planes written by the producers
// Every thread keeps ONE (i,j) for the whole march, so all array offsets are loop invariants (as in k_corner / k_final_s);
// the consumer runs one plane behind the producers and the three ring slots per array make one barrier per plane enough:
//   D = 0, 1: iteration t: A -> cA(plane k0+t), B -> cB(z-face k0+t+1)  [prologue: cB(k0)],  consumer -> faces of plane k0+t-1
//   D = 2   : iteration t: A, B -> cell plane k0+t                      [prologue: plane k0-1], consumer -> z-face k0+t-1
// Arithmetic: corner_at / final_edge, the device functions k_corner / k_final_s run, with the limited slopes read from the
// arrays k_trace wrote instead of being recomputed (same values), so the result is bit-identical to the split passes.
// SLA: the limited slopes come from the arrays k_trace wrote (true) or are recomputed from q (false: 24 B per cell and component less
// HBM traffic, the slope arithmetic again; same values)
template <bool PRED, int D, int TX, int TY, bool SLA>
__global__ void __launch_bounds__((2 * (((TX + 1) * (TY + 1) + 63) / 64) + (TX * TY + 63) / 64) * 64) k_dir(const BoxD* __restrict__ boxes,
    const FabD* __restrict__ qt, const FabD* __restrict__ ft, const FabD* __restrict__ divut,
    const FabD* __restrict__ mDt, const FabD* __restrict__ mAt, const FabD* __restrict__ mBt,
    const FabD* __restrict__ eAt, const FabD* __restrict__ eBt, const FabD* __restrict__ slDt, const FabD* __restrict__ slAt,
    const FabD* __restrict__ slBt, const FabD* __restrict__ outt, const GodParams* __restrict__ Pp,
    int ntx, int nty, int nkc, int kc, int xcd_cnt)
{
    constexpr int TA = D == 0 ? 1 : 0;
    constexpr int TB = D == 2 ? 1 : 2;
    constexpr int PW = TX + 2, PS = PW * (TY + 2);       // LDS plane with origin (tx0-1, ty0-1)
    constexpr int WP = ((TX + 1) * (TY + 1) + 63) / 64;  // wavefronts per producer role
    __shared__ double CA[3][PS], CB[3][PS];
    const GodParams& P = *Pp;
    const int fab = blockIdx.y;
    BoxD b = boxes[fab];
    b.hi[D] += 1;
    int bid = blockIdx.x;
    if (xcd_cnt > 0) {
        bid = (bid & 7) * xcd_cnt + (bid >> 3);          // XCD-aware order, see make_tiling
        if (bid >= ntx * nty * nkc) return;
    }
    const int tix = bid % ntx, r1 = bid / ntx, tiy = r1 % nty, kci = r1 / nty;
    const int tx0 = b.lo[0] + tix * TX, ty0 = b.lo[1] + tiy * TY, k0 = b.lo[2] + kci * kc;
    if (tx0 > b.hi[0] || ty0 > b.hi[1] || k0 > b.hi[2]) return;
    const int txe = min(tx0 + TX - 1, b.hi[0]), tye = min(ty0 + TY - 1, b.hi[1]), k1 = min(k0 + kc - 1, b.hi[2]);
    const int nk = k1 - k0 + 1;
    const bool has_force = P.has_force != 0, has_divu = P.has_divu != 0;
    const int n = PRED ? D : (int)blockIdx.z;
    const int cn = PRED ? 0 : n;
    const int wave = threadIdx.x >> 6;
    auto ring = [](int kk) { return ((kk % 3) + 3) % 3; };
    auto lds = [&](int i, int j) { return (i - (tx0 - 1)) + PW * (j - (ty0 - 1)); };
    if (wave < 2 * WP) {
        // ---------------- producers
        const bool isA = wave < WP;
        const int idx = isA ? (int)threadIdx.x : (int)threadIdx.x - WP * 64;
        int i0, i1, j0, j1;
        if (isA) { i0 = D == 0 ? tx0 - 1 : tx0; i1 = D == 0 ? txe : txe + 1; j0 = D == 1 ? ty0 - 1 : ty0; j1 = D == 0 ? tye + 1 : tye; }
        else { i0 = D == 0 ? tx0 - 1 : tx0; i1 = txe; j0 = D == 1 ? ty0 - 1 : ty0; j1 = D == 2 ? tye + 1 : tye; }
        const int nx = i1 - i0 + 1;
        const bool on = idx < nx * (j1 - j0 + 1);
        const int ci = i0 + idx % nx, cj = j0 + idx / nx;
        const int lo = lds(ci, cj);
        const FabD q = qt[fab], mA = mAt[fab], mB = mBt[fab];
        FabD frc; if (has_force) frc = ft[fab];
        FabD dv; if (has_divu) dv = divut[fab];
        // plane produced in iteration t: k0 + t + off (prologue: t = -1)
        const int off = (D != 2 && !isA) ? 1 : 0;
        const bool pro = D == 2 || !isA;
        if (isA) {
            const FabD eB = eBt[fab], slA = slAt[fab];
            if (pro && on) CA[ring(k0 - 1 + off)][lo] = corner_at<PRED, D, TA, SLA>(P, n, ci, cj, k0 - 1 + off, q, frc, dv, mA, mB, eB, has_force, has_divu, SLA ? &slA : nullptr);
            for (int t = 0; t <= nk; ++t) {
                if (t < nk && on) CA[ring(k0 + t + off)][lo] = corner_at<PRED, D, TA, SLA>(P, n, ci, cj, k0 + t + off, q, frc, dv, mA, mB, eB, has_force, has_divu, SLA ? &slA : nullptr);
                __syncthreads();
            }
        } else {
            const FabD eA = eAt[fab], slB = slBt[fab];
            if (pro && on) CB[ring(k0 - 1 + off)][lo] = corner_at<PRED, D, TB, SLA>(P, n, ci, cj, k0 - 1 + off, q, frc, dv, mB, mA, eA, has_force, has_divu, SLA ? &slB : nullptr);
            for (int t = 0; t <= nk; ++t) {
                if (t < nk && on) CB[ring(k0 + t + off)][lo] = corner_at<PRED, D, TB, SLA>(P, n, ci, cj, k0 + t + off, q, frc, dv, mB, mA, eA, has_force, has_divu, SLA ? &slB : nullptr);
                __syncthreads();
            }
        }
    } else {
        // ---------------- consumer
        const int idx = (int)threadIdx.x - 2 * WP * 64;
        const int fw = txe - tx0 + 1;
        const bool on = idx < fw * (tye - ty0 + 1);
        const int fi = tx0 + idx % fw, fj = ty0 + idx / fw;
        const int lo = lds(fi, fj);
        const FabD q = qt[fab], out = outt[fab], mD = mDt[fab], mA = mAt[fab], mB = mBt[fab], slD = slDt[fab];
        FabD frc; if (has_force) frc = ft[fab];
        FabD dv; if (has_divu) dv = divut[fab];
        const long mAsD = stride_of<D>(mA), mAsT = stride_of<TA>(mA), mBsD = stride_of<D>(mB), mBsT = stride_of<TB>(mB);
        for (int t = 0; t <= nk; ++t) {
            if (t > 0 && on) {
                const int k = k0 + t - 1;
                const long mAo = mA.off(fi, fj, k), mBo = mB.off(fi, fj, k);
                const double umD = mD(fi, fj, k, 0);
                const double mA_l0 = mA.gp()[mAo - mAsD], mA_l1 = mA.gp()[mAo - mAsD + mAsT], mA_h0 = mA.gp()[mAo], mA_h1 = mA.gp()[mAo + mAsT];
                const double mB_l0 = mB.gp()[mBo - mBsD], mB_l1 = mB.gp()[mBo - mBsD + mBsT], mB_h0 = mB.gp()[mBo], mB_h1 = mB.gp()[mBo + mBsT];
                const double* cAp = &CA[ring(k)][lo];
                const double* cBp = &CB[ring(k)][lo];
                long cAsD, cAsT, cBsD, cBsT;
                if (D == 0) { cAsD = 1; cAsT = PW; cBsD = 1; cBsT = (long)(&CB[ring(k + 1)][0] - &CB[ring(k)][0]); }
                else if (D == 1) { cAsD = PW; cAsT = 1; cBsD = PW; cBsT = (long)(&CB[ring(k + 1)][0] - &CB[ring(k)][0]); }
                else { cAsD = (long)(&CA[ring(k)][0] - &CA[ring(k - 1)][0]); cAsT = 1; cBsD = (long)(&CB[ring(k)][0] - &CB[ring(k - 1)][0]); cBsT = PW; }
                out(fi, fj, k, cn) = final_edge<PRED, D, SLA>(P, n, fi, fj, k, q, frc, dv, has_force, has_divu, umD, mA_l0, mA_l1, mA_h0, mA_h1,
                                                              mB_l0, mB_l1, mB_h0, mB_h1, cAp, cAsD, cAsT, cBp, cBsD, cBsT, SLA ? &slD : nullptr);
            }
            __syncthreads();
        }
    }
}

// The reconstruction is an ARGUMENT of every Godunov entry point (scheme: 0 Godunov_PLM, 1 Godunov_PPM; ns.advection_scheme,
// Source/NavierStokesBase.cpp:548-553).  The fused z-marching kernels are compiled per scheme; the multi-pass kernels read a device
// word that the entry point sets on the launch stream in front of its own launches (stream-ordered, no synchronisation): no
// process-wide mode outlives a call.
static void set_scheme(int scheme)
{
    if (scheme != 0 && scheme != 1) throw Error("iamrx Godunov: advection scheme " + std::to_string(scheme) + " is not implemented (0 Godunov_PLM, 1 Godunov_PPM, 2 BDS)");
    static const int vals[2] = {0, 1};
    static int current = -1;
    if (scheme == current) return;
    current = scheme;
    IAMRX_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ppm_dev), &vals[scheme], sizeof(int), 0, hipMemcpyHostToDevice, Context::get().stream));
}

static GodParams make_params(const Geometry& g, double dt, int ncomp, const BCRec* bc, const int* iconserv, bool is_vel, bool fit,
                             bool has_force, bool has_divu)
{
    GodParams P;
    P.dt = dt; P.ncomp = ncomp; P.is_velocity = is_vel; P.fit = fit; P.has_force = has_force; P.has_divu = has_divu;
    for (int d = 0; d < 3; ++d) { P.dx[d] = g.dx[d]; P.bc.dlo[d] = g.domain.lo[d]; P.bc.dhi[d] = g.domain.hi[d]; P.bc.per[d] = g.periodic[d]; }
    for (int n = 0; n < GOD_MAXC; ++n) {
        P.iconserv[n] = (iconserv && n < ncomp) ? iconserv[n] : 0;
        for (int d = 0; d < 3; ++d) { P.bc.bc[n].lo[d] = (bc && n < ncomp) ? bc[n].lo[d] : 0; P.bc.bc[n].hi[d] = (bc && n < ncomp) ? bc[n].hi[d] : 0; }
    }
    return P;
}

// run-time parameter block in device memory (stream-ordered ring: the H2D copy of a slot is ordered after the
// kernels that used it before; the host source is pageable, so hipMemcpyAsync stages it before returning)
static const GodParams* upload_params(const GodParams& P)
{
    static GodParams* ring = nullptr;
    static int slot = 0;
    constexpr int NSLOT = 64;
    auto& ctx = Context::get();
    if (!ring) ring = (GodParams*)ctx.alloc(NSLOT * sizeof(GodParams));
    GodParams* d = ring + (slot++ % NSLOT);
    ctx.upload_async(d, &P, sizeof(GodParams));
    return d;
}

// planes marched per thread.  Kernels whose stencil extends in z re-read the planes shared with the z-neighbouring tiles;
// those tiles are far apart in launch order, so the re-reads miss L2: march more planes per thread there.
static int tz_for(bool z_stencil)
{
    const int tzz = (int)tune("GODUNOV_TZ", 4);
    return z_stencil ? tzz : 4;
}

static bool comp_split()
{
    const int v = (int)tune("GODUNOV_COMP_SPLIT", 0);
    return v != 0;
}

static Tiling face_tiling(const Layout& l, int D, int gt, int tz)
{
    int ml[3];
    for (int e = 0; e < 3; ++e) ml[e] = l.max_len[e] + (e == D ? 1 : 2 * gt);
    return make_tiling(ml, l.nlocal(), tz);
}

template <bool PRED, int D>
static void launch_trace(const Layout& l, const MultiFab& q, const MultiFab* force, const MultiFab& mac, const MultiFab& e0, const MultiFab* sl,
                         const GodParams* dP)
{
    Tiling t = face_tiling(l, D, 1, tz_for(D == 2));
    hipLaunchKernelGGL((k_trace<PRED, D>), t.grid(), Tiling::block(), 0, Context::get().stream, t, l.d_boxes, q.d_tab,
                       force ? force->d_tab : nullptr, mac.d_tab, e0.d_tab, sl ? sl->d_tab : nullptr, dP);
}

static bool use_fused_final()
{
    const int v = tune("GODUNOV_FUSED", 0) == 1 ? 1 : 0;
    return v == 1;
}

template <bool PRED, int D, int T>
static void launch_corner(const Layout& l, const MultiFab& q, const MultiFab* force, const MultiFab* divu, const MultiFab& mT,
                          const MultiFab& mO, const MultiFab& eO, MultiFab& out, const GodParams* dP)
{
    int ml[3];
    for (int e = 0; e < 3; ++e) ml[e] = l.max_len[e] + (e == T ? 1 : 0) + (e == D ? 2 : 0);
    Tiling t = make_tiling(ml, l.nlocal(), tz_for(T == 2 || D == 2));
    dim3 gr = t.grid();
    if (!PRED && comp_split()) gr.z = (unsigned)out.ncomp;
    hipLaunchKernelGGL((k_corner<PRED, D, T>), gr, Tiling::block(), 0, Context::get().stream, t, l.d_boxes, q.d_tab,
                       force ? force->d_tab : nullptr, divu ? divu->d_tab : nullptr, mT.d_tab, mO.d_tab, eO.d_tab, out.d_tab, dP);
}

template <bool PRED, int D>
static void launch_final_split(const Layout& l, const MultiFab& q, int ncomp, const MultiFab* force, const MultiFab* divu,
                               MultiFab* const mac[3], const MultiFab e0[3], MultiFab& out, const GodParams* dP)
{
    constexpr int TA = D == 0 ? 1 : 0;
    constexpr int TB = D == 2 ? 1 : 2;
    const int nc = PRED ? 1 : ncomp;
    MultiFab cA(q.layout, face_type(TA), nc, 1), cB(q.layout, face_type(TB), nc, 1);
    launch_corner<PRED, D, TA>(l, q, force, divu, *mac[TA], *mac[TB], e0[TB], cA, dP);
    launch_corner<PRED, D, TB>(l, q, force, divu, *mac[TB], *mac[TA], e0[TA], cB, dP);
    Tiling t = face_tiling(l, D, 0, tz_for(true));
    dim3 gr = t.grid();
    if (!PRED && comp_split()) gr.z = (unsigned)nc;
    hipLaunchKernelGGL((k_final_s<PRED, D>), gr, Tiling::block(), 0, Context::get().stream, t, l.d_boxes, q.d_tab,
                       force ? force->d_tab : nullptr, divu ? divu->d_tab : nullptr, mac[D]->d_tab, mac[TA]->d_tab, mac[TB]->d_tab,
                       cA.d_tab, cB.d_tab, out.d_tab, dP);
}

static bool use_dir_fused()
{
    const int v = (int)tune("GODUNOV_DIR", 1);
    return v != 0;
}

template <bool PRED, int D, int DTX, int DTY>
static void launch_dir_t(const Layout& l, const MultiFab& q, int ncomp, const MultiFab* force, const MultiFab* divu,
                       MultiFab* const mac[3], const MultiFab e0[3], const MultiFab sl[3], MultiFab& out, const GodParams* dP)
{
    constexpr int TA = D == 0 ? 1 : 0;
    constexpr int TB = D == 2 ? 1 : 2;
    constexpr int TX = DTX, TY = DTY;
    constexpr int NT = (2 * (((TX + 1) * (TY + 1) + 63) / 64) + (TX * TY + 63) / 64) * 64;
    int nf[3];
    for (int e = 0; e < 3; ++e) nf[e] = l.max_len[e] + (e == D ? 1 : 0);
    const int ntx = (nf[0] + TX - 1) / TX, nty = (nf[1] + TY - 1) / TY;
    const int kc_env = (int)tune("GODUNOV_KC", 0);
    const int kc = kc_env > 0 ? kc_env : std::min(32, std::max(8, nf[2] / 8));       // planes marched per workgroup
    const int nkc = (nf[2] + kc - 1) / kc;
    const int total = ntx * nty * nkc;
    const int xcd_cnt = total >= 64 ? (total + 7) / 8 : 0;
    dim3 grid((unsigned)(xcd_cnt > 0 ? 8 * xcd_cnt : total), (unsigned)l.nlocal(), (unsigned)(PRED ? 1 : ncomp));
    const bool sla = tune("GODUNOV_DIR_SLOPES", 1) != 0;
#define IAMRX_KDIR(SLA) hipLaunchKernelGGL((k_dir<PRED, D, TX, TY, SLA>), grid, dim3(NT), 0, Context::get().stream, l.d_boxes, q.d_tab, \
                       force ? force->d_tab : nullptr, divu ? divu->d_tab : nullptr, mac[D]->d_tab, mac[TA]->d_tab, mac[TB]->d_tab, \
                       e0[TA].d_tab, e0[TB].d_tab, sl[D].d_tab, sl[TA].d_tab, sl[TB].d_tab, out.d_tab, dP, ntx, nty, nkc, kc, xcd_cnt)
    if (sla) IAMRX_KDIR(true); else IAMRX_KDIR(false);
#undef IAMRX_KDIR
}

template <bool PRED, int D>
static void launch_dir(const Layout& l, const MultiFab& q, int ncomp, const MultiFab* force, const MultiFab* divu,
                       MultiFab* const mac[3], const MultiFab e0[3], const MultiFab sl[3], MultiFab& out, const GodParams* dP)
{
    // 16 x 8 tiles (8 wavefronts, 2 workgroups per CU) measured 6% faster than 32 x 8 (14 wavefronts, 1 per CU) at 256^3
    const int tx = (int)tune("GODUNOV_DIR_TX", 16);
    const int ty = (int)tune("GODUNOV_DIR_TY", 8);
    if (tx == 16 && ty == 4) launch_dir_t<PRED, D, 16, 4>(l, q, ncomp, force, divu, mac, e0, sl, out, dP);
    else if (tx == 32 && ty == 4) launch_dir_t<PRED, D, 32, 4>(l, q, ncomp, force, divu, mac, e0, sl, out, dP);
    else if (tx == 16) launch_dir_t<PRED, D, 16, 8>(l, q, ncomp, force, divu, mac, e0, sl, out, dP);
    else launch_dir_t<PRED, D, 32, 8>(l, q, ncomp, force, divu, mac, e0, sl, out, dP);
}

template <bool PRED, int D>
static void launch_final(const Layout& l, const MultiFab& q, const MultiFab* force, const MultiFab* divu, MultiFab* const mac[3],
                         const MultiFab e0[3], const MultiFab sl[3], MultiFab& out, const GodParams* dP)
{
    if (use_dir_fused()) { launch_dir<PRED, D>(l, q, e0[0].ncomp, force, divu, mac, e0, sl, out, dP); return; }
    if (!use_fused_final()) { launch_final_split<PRED, D>(l, q, e0[0].ncomp, force, divu, mac, e0, out, dP); return; }
    Tiling t = face_tiling(l, D, 0, 4);
    hipLaunchKernelGGL((k_final<PRED, D>), t.grid(), Tiling::block(), 0, Context::get().stream, t, l.d_boxes, q.d_tab,
                       force ? force->d_tab : nullptr, divu ? divu->d_tab : nullptr, mac[0]->d_tab, mac[1]->d_tab, mac[2]->d_tab,
                       e0[0].d_tab, e0[1].d_tab, e0[2].d_tab, out.d_tab, dP);
}

static bool use_z_kernel();
static void godunov_pred_z(const Layout& l, const MultiFab& vel, const MultiFab* force, MultiFab* const umac[3], const GodParams* dP, bool bcs, bool ppm, const Geometry& g);
static void set_scheme(int scheme);

void godunov_extrap_vel_to_faces(const Geometry& g, const MultiFab& vel, const MultiFab* force, MultiFab* const umac[3],
                                 double dt, const BCRec* bc, bool use_forces_in_trans, int scheme)
{
    if (vel.nlocal() == 0) return;
    if (scheme == 2) scheme = 0;          // BDS: the velocity prediction is Godunov_PLM (NavierStokesBase.cpp:4487: godunov_use_ppm = (scheme == Godunov_PPM))
    set_scheme(scheme);
    IAMRX_ASSERT(vel.ngrow >= 3 && vel.ncomp >= 3);
    IAMRX_ASSERT(!force || force->ngrow >= 1);
    const Layout& l = *vel.layout;
    if (use_z_kernel()) {
        godunov_pred_z(l, vel, force, umac, upload_params(make_params(g, dt, 3, bc, nullptr, true, use_forces_in_trans, force != nullptr, false)),
                       !(g.periodic[0] && g.periodic[1] && g.periodic[2]), scheme == 1, g);
        return;
    }
    MultiFab ad[3], e0[3], sl[3];
    MultiFab* adp[3];
    for (int d = 0; d < 3; ++d) {
        ad[d].define(vel.layout, face_type(d), 1, 1); e0[d].define(vel.layout, face_type(d), 3, 1); adp[d] = &ad[d];
        if (use_dir_fused()) sl[d].define(vel.layout, cell_type(), 3, 1);
    }
    const bool ws = use_dir_fused();
    const GodParams* dP = upload_params(make_params(g, dt, 3, bc, nullptr, true, use_forces_in_trans, force != nullptr, false));
    launch_trace<true, 0>(l, vel, force, ad[0], e0[0], ws ? &sl[0] : nullptr, dP);
    launch_trace<true, 1>(l, vel, force, ad[1], e0[1], ws ? &sl[1] : nullptr, dP);
    launch_trace<true, 2>(l, vel, force, ad[2], e0[2], ws ? &sl[2] : nullptr, dP);
    launch_final<true, 0>(l, vel, force, nullptr, adp, e0, sl, *umac[0], dP);
    launch_final<true, 1>(l, vel, force, nullptr, adp, e0, sl, *umac[1], dP);
    launch_final<true, 2>(l, vel, force, nullptr, adp, e0, sl, *umac[2], dP);
}

// -------------------------------------------------------------------------------- pass 3
__global__ void __launch_bounds__(256) k_aofs(Tiling t, const BoxD* __restrict__ boxes, const FabD* __restrict__ aofst, int acomp,
    const FabD* __restrict__ ext, const FabD* __restrict__ eyt, const FabD* __restrict__ ezt,
    const FabD* __restrict__ uxt, const FabD* __restrict__ uyt, const FabD* __restrict__ uzt,
    const FabD* __restrict__ fxt, const FabD* __restrict__ fyt, const FabD* __restrict__ fzt, const GodParams* __restrict__ Pp)
{
    const GodParams& P = *Pp;
    const int fab = blockIdx.y;
    int i, j, k0, k1;
    if (!tile_ijk(t, boxes[fab], i, j, k0, k1)) return;
    const FabD aofs = aofst[fab], ex = ext[fab], ey = eyt[fab], ez = ezt[fab], ux = uxt[fab], uy = uyt[fab], uz = uzt[fab];
    const bool store_flux = fxt != nullptr;
    const double dx0 = P.dx[0], dx1 = P.dx[1], dx2 = P.dx[2];
    const double ax = dx1 * dx2, ay = dx2 * dx0, az = dx0 * dx1;
    const double qvol = 1.0 / (dx0 * dx1 * dx2);
    const int ncomp = P.ncomp;
    for (int k = k0; k <= k1; ++k) {
        const double uxl = ux(i, j, k), uxh = ux(i + 1, j, k), uyl = uy(i, j, k), uyh = uy(i, j + 1, k), uzl = uz(i, j, k), uzh = uz(i, j, k + 1);
        const double divum = 1.0 * ((uxh - uxl) / dx0 + (uyh - uyl) / dx1 + (uzh - uzl) / dx2);
        for (int n = 0; n < ncomp; ++n) {
            const double exl = ex(i, j, k, n), exh = ex(i + 1, j, k, n), eyl = ey(i, j, k, n), eyh = ey(i, j + 1, k, n), ezl = ez(i, j, k, n), ezh = ez(i, j, k + 1, n);
            const double fxl = exl * uxl * ax, fxh = exh * uxh * ax, fyl = eyl * uyl * ay, fyh = eyh * uyh * ay, fzl = ezl * uzl * az, fzh = ezh * uzh * az;
            if (store_flux) {
                fxt[fab](i, j, k, n) = fxl; fyt[fab](i, j, k, n) = fyl; fzt[fab](i, j, k, n) = fzl;
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

// -------------------------------------------------------------------------------- single-pass tile kernel (advection)
// One workgroup = one TXxTYxTZ tile of cells and one component at a time.  The state tile (3 ghost cells) is staged in
// LDS, the pass-1 transverse states E[d], the corner-coupled states C (two arrays per final direction) and the final edge
// states G[d] of the tile never leave LDS, and aofs is written straight from them: per cell and component HBM sees the
// state, the forcing, the mac velocities and aofs -- the algorithmic traffic of SURVEY 8d -- instead of 6+3 intermediate
// face arrays.  Faces on tile boundaries (and the one-cell rings the corner coupling needs) are recomputed by both tiles
// with identical arithmetic, so the result is bit-identical to the multi-pass kernels above.
struct LBox {
    int lo0, lo1, lo2, n0, n1, n2;
    __device__ __forceinline__ int off(int i, int j, int k) const { return (i - lo0) + n0 * ((j - lo1) + n1 * (k - lo2)); }
    __device__ __forceinline__ int size() const { return n0 * n1 * n2; }
    template <int D> __device__ __forceinline__ int stride() const { return D == 0 ? 1 : (D == 1 ? n0 : n0 * n1); }
    __device__ __forceinline__ void ijk(int idx, int& i, int& j, int& k) const
    {
        i = lo0 + idx % n0; const int r = idx / n0; j = lo1 + r % n1; k = lo2 + r / n1;
    }
};
__device__ __forceinline__ LBox make_lbox(const int lo[3], const int hi[3])
{
    LBox b; b.lo0 = lo[0]; b.lo1 = lo[1]; b.lo2 = lo[2]; b.n0 = hi[0] - lo[0] + 1; b.n1 = hi[1] - lo[1] + 1; b.n2 = hi[2] - lo[2] + 1; return b;
}

struct TileCtx {
    const GodParams* P;
    FabD q, frc, dv, mac[3];
    bool has_force, has_divu, fit, is_vel;
    int tlo[3], thi[3];
    LBox qb, eb[3], gb[3], mb[3], fb;
    double *Qs, *Es[3], *Gs[3], *Ca, *Cb, *Ms[3], *Fs;
};

// pass 1 on the tile: E[D] on the D-faces [tlo_D, thi_D+1] x (transverse cells grown by 1)
template <int D>
__device__ __forceinline__ void tile_stage1(const TileCtx& c, int n)
{
    const GodParams& P = *c.P;
    const LBox eb = c.eb[D];
    const int s = c.qb.template stride<D>();
    const double hdt = 0.5 * P.dt, dtdx = P.dt / P.dx[D];
    const int domlo = P.bc.dlo[D], domhi = P.bc.dhi[D];
    const bool nonper = !P.bc.per[D];
    const int bl = P.bc.bc[n].lo[D], bh = P.bc.bc[n].hi[D];
    const bool edlo = nonper && ed_or_ho(bl), edhi = nonper && ed_or_ho(bh);
    const int fs = c.fb.template stride<D>();
    for (int idx = threadIdx.x; idx < eb.size(); idx += blockDim.x) {
        int i, j, k; eb.ijk(idx, i, j, k);
        const int f = D == 0 ? i : (D == 1 ? j : k);
        const double uad = c.Ms[D][c.mb[D].off(i, j, k)];
        const double fu = (fabs(uad) < SMALL_VEL) ? 0.0 : 1.0;
        const double* qn = c.Qs + c.qb.off(i, j, k);
        double l, h;
        trace_lohi<false>(qn, nullptr, s, uad, dtdx, edlo, edhi, f, domlo, domhi, l, h);
        if (c.fit && c.has_force) { const int fo = c.fb.off(i, j, k); l += hdt * c.Fs[fo - fs]; h += hdt * c.Fs[fo]; }
        if (nonper) trans_bc(qn, s, f, c.is_vel && n == D, l, h, bl, bh, domlo, domhi);
        const double st = (uad >= 0.) ? l : h;
        c.Es[D][idx] = fu * st + (1.0 - fu) * 0.5 * (h + l);
    }
}

// pass 2a on the tile: corner-coupled states on the T-faces [tlo_T, thi_T+1], D cells grown by 1, O cells of the tile
template <int D, int T>
__device__ __forceinline__ void tile_corner(const TileCtx& c, int n, double* out, const LBox& cb)
{
    constexpr int O = 3 - D - T;
    const GodParams& P = *c.P;
    const bool conserv = P.iconserv[n] != 0;
    const double c_o = conserv ? P.dt / (3.0 * P.dx[O]) : P.dt / (6.0 * P.dx[O]);
    const double dt3 = P.dt / 3.0, dtdxT = P.dt / P.dx[T], hdt = 0.5 * P.dt;
    const bool nonperT = !P.bc.per[T];
    const int dlo = P.bc.dlo[T], dhi = P.bc.dhi[T];
    const int qsT = c.qb.template stride<T>();
    const LBox eo = c.eb[O];
    const int eOsT = eo.template stride<T>(), eOsO = eo.template stride<O>();
    const LBox mo = c.mb[O];
    const int mOsT = mo.template stride<T>(), mOsO = mo.template stride<O>();
    const int fsT = c.fb.template stride<T>();
    const long dsT = c.has_divu ? stride_of<T>(c.dv) : 0;
    for (int idx = threadIdx.x; idx < cb.size(); idx += blockDim.x) {
        int i, j, k; cb.ijk(idx, i, j, k);
        const int fT = T == 0 ? i : (T == 1 ? j : k);
        const double macT = c.Ms[T][c.mb[T].off(i, j, k)];
        out[idx] = corner_state<false>(c.Qs + c.qb.off(i, j, k), nullptr, qsT, fT, macT, c.Ms[O] + mo.off(i, j, k), mOsT, mOsO,
            c.Es[O] + eo.off(i, j, k), eOsT, eOsO, c.has_force ? c.Fs + c.fb.off(i, j, k) : nullptr, fsT, dtdxT, c_o, dt3,
            P.dx[O], conserv, c.has_divu ? c.dv.p + c.dv.off(i, j, k) : nullptr, dsT, c.fit, hdt, nonperT, c.is_vel && n == T,
            P.bc.bc[n].lo[T], P.bc.bc[n].hi[T], dlo, dhi);
    }
}

// pass 2b on the tile: final edge states on the D-faces [tlo_D, thi_D+1] of the tile's cells -> G[D]
template <int D>
__device__ __forceinline__ void tile_final(const TileCtx& c, int n, const LBox& ab, const LBox& bb)
{
    constexpr int TA = D == 0 ? 1 : 0;
    constexpr int TB = D == 2 ? 1 : 2;
    const GodParams& P = *c.P;
    const LBox gb = c.gb[D];
    const int qsD = c.qb.template stride<D>();
    const int fsD = c.fb.template stride<D>();
    const long dsD = c.has_divu ? stride_of<D>(c.dv) : 0;
    const LBox ma = c.mb[TA], mbx = c.mb[TB];
    const double* mAp = c.Ms[TA]; const double* mBp = c.Ms[TB];
    const int mAsD = ma.template stride<D>(), mAsT = ma.template stride<TA>(), mBsD = mbx.template stride<D>(), mBsT = mbx.template stride<TB>();
    const int cAsD = ab.template stride<D>(), cAsT = ab.template stride<TA>(), cBsD = bb.template stride<D>(), cBsT = bb.template stride<TB>();
    const double dt = P.dt, hdt = 0.5 * P.dt, dtdxD = dt / P.dx[D];
    const bool nonperD = !P.bc.per[D];
    const int dloD = P.bc.dlo[D], dhiD = P.bc.dhi[D];
    const bool conserv = P.iconserv[n] != 0;
    const int blD = P.bc.bc[n].lo[D], bhD = P.bc.bc[n].hi[D];
    const bool edlo = nonperD && ed_or_ho(blD), edhi = nonperD && ed_or_ho(bhD);
    for (int idx = threadIdx.x; idx < gb.size(); idx += blockDim.x) {
        int i, j, k; gb.ijk(idx, i, j, k);
        const int f = D == 0 ? i : (D == 1 ? j : k);
        const double* qn = c.Qs + c.qb.off(i, j, k);
        const int fo = c.fb.off(i, j, k);
        const long dvo = c.has_divu ? c.dv.off(i, j, k) : 0;
        const int mAo = ma.off(i, j, k), mBo = mbx.off(i, j, k);
        const double umD = c.Ms[D][c.mb[D].off(i, j, k)];
        const double mA_l0 = mAp[mAo - mAsD], mA_l1 = mAp[mAo - mAsD + mAsT], mA_h0 = mAp[mAo], mA_h1 = mAp[mAo + mAsT];
        const double mB_l0 = mBp[mBo - mBsD], mB_l1 = mBp[mBo - mBsD + mBsT], mB_h0 = mBp[mBo], mB_h1 = mBp[mBo + mBsT];
        double stl, sth;
        trace_lohi<false>(qn, nullptr, qsD, umD, dtdxD, edlo, edhi, f, dloD, dhiD, stl, sth);
        if (c.fit && c.has_force) { stl += hdt * c.Fs[fo - fsD]; sth += hdt * c.Fs[fo]; }
        if (nonperD) trans_bc(qn, qsD, f, c.is_vel && n == D, stl, sth, blD, bhD, dloD, dhiD);
        const int cAo = ab.off(i, j, k), cBo = bb.off(i, j, k);
        const double Al0 = c.Ca[cAo - cAsD], Al1 = c.Ca[cAo - cAsD + cAsT], Ah0 = c.Ca[cAo], Ah1 = c.Ca[cAo + cAsT];
        const double Bl0 = c.Cb[cBo - cBsD], Bl1 = c.Cb[cBo - cBsD + cBsT], Bh0 = c.Cb[cBo], Bh1 = c.Cb[cBo + cBsT];
        if (conserv) {
            const double cfA = 0.5 * dt / P.dx[TA], cfB = 0.5 * dt / P.dx[TB];
            stl += -cfA * (Al1 * mA_l1 - Al0 * mA_l0);
            sth += -cfA * (Ah1 * mA_h1 - Ah0 * mA_h0);
            stl += -cfB * (Bl1 * mB_l1 - Bl0 * mB_l0);
            sth += -cfB * (Bh1 * mB_h1 - Bh0 * mB_h0);
            stl += cfA * qn[-qsD] * (mA_l1 - mA_l0);
            sth += cfA * qn[0] * (mA_h1 - mA_h0);
            stl += cfB * qn[-qsD] * (mB_l1 - mB_l0);
            sth += cfB * qn[0] * (mB_h1 - mB_h0);
            if (c.has_divu) { stl -= 0.5 * dt * qn[-qsD] * c.dv.p[dvo - dsD]; sth -= 0.5 * dt * qn[0] * c.dv.p[dvo]; }
        } else {
            const double cfA = 0.25 * dt / P.dx[TA], cfB = 0.25 * dt / P.dx[TB];
            stl -= cfA * (mA_l1 + mA_l0) * (Al1 - Al0);
            sth -= cfA * (mA_h1 + mA_h0) * (Ah1 - Ah0);
            stl -= cfB * (mB_l1 + mB_l0) * (Bl1 - Bl0);
            sth -= cfB * (mB_h1 + mB_h0) * (Bh1 - Bh0);
        }
        if (!c.fit && c.has_force) { stl += hdt * c.Fs[fo - fsD]; sth += hdt * c.Fs[fo]; }
        if (nonperD) edge_bc(qn, qsD, f, c.is_vel && n == D, stl, sth, blD, bhD, dloD, dhiD);
        double temp = (umD >= 0.) ? stl : sth;
        temp = (fabs(umD) < SMALL_VEL) ? 0.5 * (stl + sth) : temp;
        c.Gs[D][idx] = temp;
    }
}

// both corner arrays + the final states of direction D
template <int D>
__device__ __forceinline__ void tile_direction(const TileCtx& c, int n)
{
    constexpr int TA = D == 0 ? 1 : 0;
    constexpr int TB = D == 2 ? 1 : 2;
    int lo[3], hi[3];
    for (int e = 0; e < 3; ++e) { lo[e] = c.tlo[e]; hi[e] = c.thi[e]; }
    lo[D] -= 1; hi[D] += 1;
    int la[3] = {lo[0], lo[1], lo[2]}, ha[3] = {hi[0], hi[1], hi[2]};
    ha[TA] += 1;
    const LBox ab = make_lbox(la, ha);
    int lb[3] = {lo[0], lo[1], lo[2]}, hb[3] = {hi[0], hi[1], hi[2]};
    hb[TB] += 1;
    const LBox bb = make_lbox(lb, hb);
    tile_corner<D, TA>(c, n, c.Ca, ab);
    tile_corner<D, TB>(c, n, c.Cb, bb);
    __syncthreads();
    tile_final<D>(c, n, ab, bb);
    __syncthreads();
}

template <int TX, int TY, int TZ, int NTH>
__global__ void __launch_bounds__(NTH) k_godunov_tile(const BoxD* __restrict__ boxes, const FabD* __restrict__ qt, const FabD* __restrict__ ft,
    const FabD* __restrict__ divut, const FabD* __restrict__ uxt, const FabD* __restrict__ uyt, const FabD* __restrict__ uzt,
    const FabD* __restrict__ aofst, int acomp, const FabD* __restrict__ e0t, const FabD* __restrict__ e1t, const FabD* __restrict__ e2t,
    const FabD* __restrict__ f0t, const FabD* __restrict__ f1t, const FabD* __restrict__ f2t,
    const GodParams* __restrict__ Pp, int ntx, int nty, int ntz)
{
    constexpr int NQ = (TX + 6) * (TY + 6) * (TZ + 6);
    constexpr int NE0 = (TX + 1) * (TY + 2) * (TZ + 2), NE1 = (TX + 2) * (TY + 1) * (TZ + 2), NE2 = (TX + 2) * (TY + 2) * (TZ + 1);
    constexpr int NG0 = (TX + 1) * TY * TZ, NG1 = TX * (TY + 1) * TZ, NG2 = TX * TY * (TZ + 1);
    constexpr int c01 = (TX + 2) * (TY + 1) * TZ, c02 = (TX + 2) * TY * (TZ + 1);     // D = 0: T = 1, 2
    constexpr int c10 = (TX + 1) * (TY + 2) * TZ, c12 = TX * (TY + 2) * (TZ + 1);     // D = 1: T = 0, 2
    constexpr int c20 = (TX + 1) * TY * (TZ + 2), c21 = TX * (TY + 1) * (TZ + 2);     // D = 2: T = 0, 1
    constexpr int NCA = c01 > c10 ? (c01 > c20 ? c01 : c20) : (c10 > c20 ? c10 : c20);
    constexpr int NCB = c02 > c12 ? (c02 > c21 ? c02 : c21) : (c12 > c21 ? c12 : c21);
    __shared__ double Qs[NQ];
    __shared__ double E0[NE0], E1[NE1], E2[NE2];
    __shared__ double G0[NG0], G1[NG1], G2[NG2];
    __shared__ double Ca[NCA], Cb[NCB];
    constexpr int NM0 = (TX + 3) * (TY + 2) * (TZ + 2), NM1 = (TX + 2) * (TY + 3) * (TZ + 2), NM2 = (TX + 2) * (TY + 2) * (TZ + 3);
    constexpr int NF = (TX + 2) * (TY + 2) * (TZ + 2);
    __shared__ double M0[NM0], M1[NM1], M2[NM2];
    __shared__ double Fs[NF];
    const GodParams& P = *Pp;
    const int fab = blockIdx.y;
    const BoxD bx = boxes[fab];
    const int bid = blockIdx.x;
    const int tix = bid % ntx, r1 = bid / ntx, tiy = r1 % nty, tiz = r1 / nty;
    TileCtx c;
    c.P = Pp;
    c.tlo[0] = bx.lo[0] + tix * TX; c.tlo[1] = bx.lo[1] + tiy * TY; c.tlo[2] = bx.lo[2] + tiz * TZ;
    if (c.tlo[0] > bx.hi[0] || c.tlo[1] > bx.hi[1] || c.tlo[2] > bx.hi[2]) return;
    c.thi[0] = min(c.tlo[0] + TX - 1, bx.hi[0]); c.thi[1] = min(c.tlo[1] + TY - 1, bx.hi[1]); c.thi[2] = min(c.tlo[2] + TZ - 1, bx.hi[2]);
    c.q = qt[fab];
    c.has_force = P.has_force != 0; c.has_divu = P.has_divu != 0; c.fit = P.fit != 0; c.is_vel = P.is_velocity != 0;
    if (c.has_force) c.frc = ft[fab];
    if (c.has_divu) c.dv = divut[fab];
    c.mac[0] = uxt[fab]; c.mac[1] = uyt[fab]; c.mac[2] = uzt[fab];
    c.Ms[0] = M0; c.Ms[1] = M1; c.Ms[2] = M2; c.Fs = Fs;
    c.Qs = Qs; c.Es[0] = E0; c.Es[1] = E1; c.Es[2] = E2; c.Gs[0] = G0; c.Gs[1] = G1; c.Gs[2] = G2; c.Ca = Ca; c.Cb = Cb;
    {
        int lo[3], hi[3];
        for (int e = 0; e < 3; ++e) { lo[e] = c.tlo[e] - 3; hi[e] = c.thi[e] + 3; }
        c.qb = make_lbox(lo, hi);
        for (int d = 0; d < 3; ++d) {
            for (int e = 0; e < 3; ++e) { lo[e] = c.tlo[e] - (e == d ? 0 : 1); hi[e] = c.thi[e] + 1; }
            c.eb[d] = make_lbox(lo, hi);
            for (int e = 0; e < 3; ++e) { lo[e] = c.tlo[e]; hi[e] = c.thi[e] + (e == d ? 1 : 0); }
            c.gb[d] = make_lbox(lo, hi);
            for (int e = 0; e < 3; ++e) { lo[e] = c.tlo[e] - 1; hi[e] = c.thi[e] + 1 + (e == d ? 1 : 0); }
            c.mb[d] = make_lbox(lo, hi);
        }
        for (int e = 0; e < 3; ++e) { lo[e] = c.tlo[e] - 1; hi[e] = c.thi[e] + 1; }
        c.fb = make_lbox(lo, hi);
    }
    // the mac velocities of the tile (cells grown by 1) are shared by all components
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const FabD::gdouble* mp = (const FabD::gdouble*)c.mac[d].p;
        for (int idx = threadIdx.x; idx < c.mb[d].size(); idx += NTH) {
            int i, j, k; c.mb[d].ijk(idx, i, j, k);
            c.Ms[d][idx] = mp[c.mac[d].off(i, j, k)];
        }
    }
    const FabD aofs = aofst[fab];
    const double dx0 = P.dx[0], dx1 = P.dx[1], dx2 = P.dx[2];
    const double ax = dx1 * dx2, ay = dx2 * dx0, az = dx0 * dx1;
    const double qvol = 1.0 / (dx0 * dx1 * dx2);
    const int ncomp = P.ncomp;
    const int tnx = c.thi[0] - c.tlo[0] + 1, tny = c.thi[1] - c.tlo[1] + 1, tnz = c.thi[2] - c.tlo[2] + 1;
    for (int n = 0; n < ncomp; ++n) {
        // stage the state tile (3 ghost cells) of component n
        {
            const FabD::gdouble* qp = (const FabD::gdouble*)c.q.p + c.q.cs * n;
            for (int idx = threadIdx.x; idx < c.qb.size(); idx += NTH) {
                int i, j, k; c.qb.ijk(idx, i, j, k);
                Qs[idx] = qp[c.q.off(i, j, k)];
            }
            if (c.has_force) {
                const FabD::gdouble* fp = (const FabD::gdouble*)c.frc.p + c.frc.cs * n;
                for (int idx = threadIdx.x; idx < c.fb.size(); idx += NTH) {
                    int i, j, k; c.fb.ijk(idx, i, j, k);
                    Fs[idx] = fp[c.frc.off(i, j, k)];
                }
            }
        }
        __syncthreads();
        tile_stage1<0>(c, n);
        tile_stage1<1>(c, n);
        tile_stage1<2>(c, n);
        __syncthreads();
        tile_direction<0>(c, n);
        tile_direction<1>(c, n);
        tile_direction<2>(c, n);
        // aofs (ComputeFluxes, ComputeDivergence(-1), ComputeConvectiveTerm) and the optional edge-state / flux outputs
        for (int idx = threadIdx.x; idx < tnx * tny * tnz; idx += NTH) {
            const int i = c.tlo[0] + idx % tnx, r = idx / tnx, j = c.tlo[1] + r % tny, k = c.tlo[2] + r / tny;
            const double uxl = M0[c.mb[0].off(i, j, k)], uxh = M0[c.mb[0].off(i + 1, j, k)], uyl = M1[c.mb[1].off(i, j, k)], uyh = M1[c.mb[1].off(i, j + 1, k)];
            const double uzl = M2[c.mb[2].off(i, j, k)], uzh = M2[c.mb[2].off(i, j, k + 1)];
            const double exl = G0[c.gb[0].off(i, j, k)], exh = G0[c.gb[0].off(i + 1, j, k)];
            const double eyl = G1[c.gb[1].off(i, j, k)], eyh = G1[c.gb[1].off(i, j + 1, k)];
            const double ezl = G2[c.gb[2].off(i, j, k)], ezh = G2[c.gb[2].off(i, j, k + 1)];
            const double fxl = exl * uxl * ax, fxh = exh * uxh * ax, fyl = eyl * uyl * ay, fyh = eyh * uyh * ay, fzl = ezl * uzl * az, fzh = ezh * uzh * az;
            if (e0t) {
                e0t[fab](i, j, k, n) = exl; e1t[fab](i, j, k, n) = eyl; e2t[fab](i, j, k, n) = ezl;
                if (i == bx.hi[0]) e0t[fab](i + 1, j, k, n) = exh;
                if (j == bx.hi[1]) e1t[fab](i, j + 1, k, n) = eyh;
                if (k == bx.hi[2]) e2t[fab](i, j, k + 1, n) = ezh;
            }
            if (f0t) {
                f0t[fab](i, j, k, n) = fxl; f1t[fab](i, j, k, n) = fyl; f2t[fab](i, j, k, n) = fzl;
                if (i == bx.hi[0]) f0t[fab](i + 1, j, k, n) = fxh;
                if (j == bx.hi[1]) f1t[fab](i, j + 1, k, n) = fyh;
                if (k == bx.hi[2]) f2t[fab](i, j, k + 1, n) = fzh;
            }
            double upd = -1.0 * qvol * ((fxh - fxl) + (fyh - fyl) + (fzh - fzl));
            if (!P.iconserv[n]) {
                const double divum = 1.0 * ((uxh - uxl) / dx0 + (uyh - uyl) / dx1 + (uzh - uzl) / dx2);
                double qavg = exl + exh + eyl + eyh + ezl + ezh;
                qavg *= 1.0 / 6.0;
                upd += qavg * divum;
            }
            aofs(i, j, k, acomp + n) = -upd;
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------- fused z-marching advection (default, PLM)
// k_god_z: one launch = ComputeFluxesOnBoxFromState + ComputeDivergence / ComputeConvectiveTerm for one component per blockIdx.z.
// A workgroup owns a TX x TY column of cells and marches through the planes of its z-chunk; thread (ci,cj) of the tile grown by one
// cell keeps its column for the whole march: the z-stencil (q(P-2..P+2), the z-slopes, the traced z-states) lives in registers, the
// in-plane stencils read one state plane (3 ghost cells) from LDS, and every intermediate of the corner-transport scheme -- the
// pass-1 states E_d, the six corner-coupled states C_{T|O} and the final edge states -- passes through two-slot LDS ring planes and
// never reaches HBM.  Iteration P (k0-1 .. k1+1):
//   stage 0  prefetched registers -> LDS: state plane P, mac_x / mac_y / force / divu of plane P, mac_z of face P
//   stage A  E_x(P), E_y(P), E_z(face P)            (limited slopes of the own and of the low-side neighbour cell, trace, BCs, upwind)
//   stage B  C_{x|y}(P), C_{y|x}(P), C_{z|x}(P), C_{z|y}(P)  and  C_{x|z}(P-1), C_{y|z}(P-1)   (these need E_z of the faces P-1, P)
//   stage C  final states of the x- and y-faces of plane P-1 and of the z-face P
//   stage D  fluxes and aofs of plane P-1 (the x / y states of the high faces come from the neighbour threads through LDS)
// Four barriers per plane; all array offsets of a thread are loop invariants.  HBM sees the state (1.6x for the in-plane halo + the
// own column), mac, forcing, divu and aofs.  Arithmetic: slope4v / trans_bc_v / corner_core / final_core = the expressions of the
// multi-pass kernels above, which remain as the PPM path and as the reference of tests/test_gpu_godunov_fused.py.
struct GodTabs3 { const FabD* t[3]; };
// z-chunks of the fused kernels: chunk c of a box covers the planes lo + start[c] .. lo + start[c + 1] - 1.  Uniform chunks, except on a
// level with domain walls in z, where a thin first and last chunk keep the boundary-condition variant of the kernels to a few planes.
struct GodChunks { int n; int start[19]; };
static GodChunks god_chunks(int len, int kc, bool thin_ends)
{
    GodChunks c;
    c.n = 0; c.start[0] = 0;
    constexpr int THIN = 8;
    int lo = 0, hi = len;
    if (thin_ends && len >= 4 * THIN) { c.start[++c.n] = THIN; lo = THIN; hi = len - THIN; }
    const int nmid = std::max(1, std::min(16, (hi - lo + kc - 1) / kc));
    for (int i = 1; i <= nmid; ++i) c.start[++c.n] = lo + (int)(((long)(hi - lo) * i) / nmid);
    if (hi < len) c.start[++c.n] = len;
    return c;
}
// A domain with walls: the tiles no boundary condition reaches run the BCS = false code in a launch of their own, the boundary-condition
// variant takes the rest.  Each launch works through a compact list of its tiles (fab, tile x, tile y, z-chunk) -- a launch over the full
// tile grid that lets the other class return at once leaves whole XCDs idle (the thin z-chunks and the first / last tile rows fall on
// XCD 0 and 7 in the XCD-aware order: measured 3.83 + 1.89 ms against 5.19 ms for the one-launch wall variant at 256^3).
// Boundary-condition code looks at the cells dlo, dlo + 1, dhi - 1, dhi (slopes) and the faces dlo, dhi + 1; a tile forms slopes of the
// cells tx0 - 2 .. txe + 1 and states of the faces tx0 - 1 .. txe + 1 (likewise in y; planes k0 - 1 .. k1 + 1); one cell of margin.
struct GodTileList { const int4* d[2]; int n[2]; int xcd_cnt[2]; };
static bool god_tile_at_wall(const Geometry& g, int tx0, int txe, int ty0, int tye, int k0, int k1)
{
    bool w = false;
    if (!g.periodic[0]) w = w || tx0 <= g.domain.lo[0] + 4 || txe >= g.domain.hi[0] - 3;
    if (!g.periodic[1]) w = w || ty0 <= g.domain.lo[1] + 4 || tye >= g.domain.hi[1] - 3;
    if (!g.periodic[2]) w = w || k0 <= g.domain.lo[2] + 3 || k1 >= g.domain.hi[2] - 3;
    return w;
}
// lists [0]: tiles off the walls, [1]: tiles at the walls.  The workgroups b, b + 8, b + 16, ... of a launch run on XCD b % 8 and take the
// entries [q, q + 1) * xcd_cnt of the list (q = b % 8): the tiles are dealt to eight buckets in runs of neighbouring tiles (shared halos meet in
// one L2), each run to the bucket with the least work so far (work of a tile = the planes it marches + its prologue; the wall list mixes
// tiles of 8 and of ~60 planes), the longest tiles first inside a bucket; short buckets are padded with dead entries (fab < 0).
// Cached per (layout, tile shape, chunks, domain).
static GodTileList god_tile_lists(const Layout& l, const Geometry& g, int TX, int TY, const GodChunks& zc)
{
    struct Entry { std::vector<long> key; int4* d[2]; int n[2]; int xc[2]; };
    static std::vector<Entry> cache;
    // entries leave with their layout (a regridding run on a domain with walls makes a new layout per regrid: ADVICE round 5) ...
    static const bool registered = [] {
        register_layout_evictor([](uint64_t lid) {
            bool any = false;
            for (const Entry& e : cache) any = any || (uint64_t)e.key[0] == lid;
            if (!any) return;
            Context::get().sync();
            for (auto it = cache.begin(); it != cache.end();) {
                if ((uint64_t)it->key[0] != lid) { ++it; continue; }
                for (int c = 0; c < 2; ++c) if (it->d[c]) (void)hipFree(it->d[c]);
                it = cache.erase(it);
            }
        });
        return true;
    }();
    (void)registered;
    std::vector<long> key = {(long)l.id, TX, TY, zc.n};
    for (int i = 0; i <= zc.n; ++i) key.push_back(zc.start[i]);
    for (int d = 0; d < 3; ++d) { key.push_back(g.domain.lo[d]); key.push_back(g.domain.hi[d]); key.push_back(g.periodic[d]); }
    for (const Entry& e : cache) if (e.key == key) return GodTileList{{e.d[0], e.d[1]}, {e.n[0], e.n[1]}, {e.xc[0], e.xc[1]}};
    if (cache.size() >= 64) {                    // ... (fallback: other tile shapes / chunkings of living layouts)
        Context::get().sync();
        for (Entry& e : cache) for (int c = 0; c < 2; ++c) if (e.d[c]) IAMRX_HIP_CHECK(hipFree(e.d[c]));
        cache.clear();
    }
    struct T { int4 t; int w; };
    std::vector<T> h[2];
    for (int f = 0; f < l.nlocal(); ++f) {
        const BoxD& b = l.lbox(f);
        const int ntx = (b.len(0) + TX - 1) / TX, nty = (b.len(1) + TY - 1) / TY;
        for (int c = 0; c < zc.n; ++c) {
            const int k0 = b.lo[2] + zc.start[c], k1 = std::min(b.lo[2] + zc.start[c + 1] - 1, b.hi[2]);
            if (k0 > b.hi[2]) break;
            for (int ty = 0; ty < nty; ++ty)
                for (int tx = 0; tx < ntx; ++tx) {
                    const int tx0 = b.lo[0] + tx * TX, ty0 = b.lo[1] + ty * TY;
                    const int txe = std::min(tx0 + TX - 1, b.hi[0]), tye = std::min(ty0 + TY - 1, b.hi[1]);
                    h[god_tile_at_wall(g, tx0, txe, ty0, tye, k0, k1) ? 1 : 0].push_back(T{make_int4(f, tx, ty, c), (k1 - k0 + 1) + 6});
                }
        }
    }
    Entry e;
    e.key = key;
    for (int c = 0; c < 2; ++c) {
        std::vector<int4> list;
        e.xc[c] = 0;
        if (h[c].size() < 64) for (const T& t : h[c]) list.push_back(t.t);
        else {
            std::vector<T> bucket[8];
            long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            constexpr size_t RUN = 6;
            for (size_t i = 0; i < h[c].size(); i += RUN) {
                int q = 0;
                for (int r = 1; r < 8; ++r) if (load[r] < load[q]) q = r;
                for (size_t j = i; j < std::min(i + RUN, h[c].size()); ++j) { bucket[q].push_back(h[c][j]); load[q] += h[c][j].w; }
            }
            size_t mx = 0;
            for (int q = 0; q < 8; ++q) {
                std::stable_sort(bucket[q].begin(), bucket[q].end(), [](const T& a, const T& b) { return a.w > b.w; });
                mx = std::max(mx, bucket[q].size());
            }
            for (int q = 0; q < 8; ++q)
                for (size_t i = 0; i < mx; ++i) list.push_back(i < bucket[q].size() ? bucket[q][i].t : make_int4(-1, 0, 0, 0));
            e.xc[c] = (int)mx;
        }
        e.n[c] = (int)list.size(); e.d[c] = nullptr;
        if (e.n[c] > 0) {
            IAMRX_HIP_CHECK(hipMalloc(&e.d[c], list.size() * sizeof(int4)));
            IAMRX_HIP_CHECK(hipMemcpy(e.d[c], list.data(), list.size() * sizeof(int4), hipMemcpyHostToDevice));
        }
    }
    cache.push_back(e);
    return GodTileList{{e.d[0], e.d[1]}, {e.n[0], e.n[1]}, {e.xc[0], e.xc[1]}};
}
template <int TX, int TY, int NT, int WPE, bool BCS, bool PPM>
__global__ void __launch_bounds__(NT, WPE) k_god_z(const BoxD* __restrict__ boxes, const FabD* __restrict__ qt, const FabD* __restrict__ ft,
    const FabD* __restrict__ divut, const FabD* __restrict__ uxt, const FabD* __restrict__ uyt, const FabD* __restrict__ uzt,
    const FabD* __restrict__ aofst, int acomp, GodTabs3 edge_t, GodTabs3 flux_t, const GodParams* __restrict__ Pp,
    int ntx, int nty, GodChunks zc, const int4* __restrict__ tiles, int ntiles, int xcd_cnt)
{
    constexpr int PW = TX + 2, PH = TY + 2, PS = PW * PH, QW = TX + 6, QH = TY + 6, QS = QW * QH;
    constexpr int NQ = (QS + NT - 1) / NT;
    static_assert(NT >= PS, "one thread per column of the grown tile");
    __shared__ double Qb[QS + QW];      // one spare row in front: the (unused) slope of the cell below the grown tile reads it
    __shared__ double MX[2][PS], MY[2][PS], MZ[2][PS], FR[2][PS], DV[2][PS];
    __shared__ double EX[2][PS], EY[2][PS], EZ[2][PS];
    __shared__ double CZX[2][PS], CZY[2][PS], CXY[2][PS], CYX[2][PS], CXZ[PS], CYZ[PS];
    double* const Q = Qb + QW;
    const GodParams& P = *Pp;
    int bid = blockIdx.x;
    if (xcd_cnt > 0) {
        bid = (bid & 7) * xcd_cnt + (bid >> 3);          // XCD-aware order, see make_tiling
        if (bid >= (tiles ? ntiles : ntx * nty * zc.n)) return;
    }
    // the tile: from the launch's list (god_tile_lists) or from the full tile grid of the box blockIdx.y
    int fab = blockIdx.y, tix, tiy, kci;
    if (tiles) { const int4 t = tiles[bid]; if (t.x < 0) return; fab = t.x; tix = t.y; tiy = t.z; kci = t.w; }
    else { tix = bid % ntx; const int r1 = bid / ntx; tiy = r1 % nty; kci = r1 / nty; }
    const BoxD b = boxes[fab];
    const int tx0 = b.lo[0] + tix * TX, ty0 = b.lo[1] + tiy * TY, k0 = b.lo[2] + zc.start[kci];
    if (tx0 > b.hi[0] || ty0 > b.hi[1] || k0 > b.hi[2]) return;
    const int txe = min(tx0 + TX - 1, b.hi[0]), tye = min(ty0 + TY - 1, b.hi[1]), k1 = min(b.lo[2] + zc.start[kci + 1] - 1, b.hi[2]);
    const int n = blockIdx.z;
    const int tid = threadIdx.x;
    const int li = tid % PW, lj = tid / PW;
    const bool act = tid < PS && tx0 - 1 + li <= txe + 1 && ty0 - 1 + lj <= tye + 1;
    const int ci = act ? tx0 - 1 + li : tx0, cj = act ? ty0 - 1 + lj : ty0;
    // LDS offsets of the own column and of its neighbours (clamped inside the grown tile: values read through a clamped offset
    // only reach results that are not used)
    const int o = (ci - (tx0 - 1)) + PW * (cj - (ty0 - 1));
    const int oxm = ci > tx0 - 1 ? o - 1 : o, oxp = ci < txe + 1 ? o + 1 : o;
    const int oym = cj > ty0 - 1 ? o - PW : o, oyp = cj < tye + 1 ? o + PW : o;
    const int oxm_yp = oxm + (oyp - o), oxp_ym = oxp + (oym - o);
    const int qo = (ci - (tx0 - 3)) + QW * (cj - (ty0 - 3));
    const bool in_tile = act && ci >= tx0 && ci <= txe && cj >= ty0 && cj <= tye;

    const bool has_force = P.has_force != 0, has_divu = P.has_divu != 0, fit = P.fit != 0, is_vel = P.is_velocity != 0;
    const bool early_force = fit && has_force, late_force = !fit && has_force;
    const bool conserv = P.iconserv[n] != 0;
    const double dt = P.dt, hdt = 0.5 * dt, dt3 = dt / 3.0;
    const double dx0 = P.dx[0], dx1 = P.dx[1], dx2 = P.dx[2];
    const double dtdx0 = dt / dx0, dtdx1 = dt / dx1, dtdx2 = dt / dx2;
    const double co0 = conserv ? dt / (3.0 * dx0) : dt / (6.0 * dx0), co1 = conserv ? dt / (3.0 * dx1) : dt / (6.0 * dx1),
                 co2 = conserv ? dt / (3.0 * dx2) : dt / (6.0 * dx2);
    const bool np0 = BCS && !P.bc.per[0], np1 = BCS && !P.bc.per[1], np2 = BCS && !P.bc.per[2];   // BCS = false: periodic in every direction
    const int bl0 = P.bc.bc[n].lo[0], bh0 = P.bc.bc[n].hi[0], bl1 = P.bc.bc[n].lo[1], bh1 = P.bc.bc[n].hi[1],
              bl2 = P.bc.bc[n].lo[2], bh2 = P.bc.bc[n].hi[2];
    const int dl0 = P.bc.dlo[0], dh0 = P.bc.dhi[0], dl1 = P.bc.dlo[1], dh1 = P.bc.dhi[1], dl2 = P.bc.dlo[2], dh2 = P.bc.dhi[2];
    const bool edl0 = np0 && ed_or_ho(bl0), edh0 = np0 && ed_or_ho(bh0), edl1 = np1 && ed_or_ho(bl1), edh1 = np1 && ed_or_ho(bh1),
               edl2 = np2 && ed_or_ho(bl2), edh2 = np2 && ed_or_ho(bh2);
    const bool nv0 = is_vel && n == 0, nv1 = is_vel && n == 1, nv2 = is_vel && n == 2;

    // global pointers of the own column at plane / face k0-1 (state: plane k0+1 = the next one the z-ring takes in)
    const FabD q = qt[fab], ux = uxt[fab], uy = uyt[fab], uz = uzt[fab], aofs = aofst[fab];
    FabD frc = q, dv = q;
    if (has_force) frc = ft[fab];
    if (has_divu) dv = divut[fab];
    const long qsz = (long)q.n[0] * q.n[1], uxsz = (long)ux.n[0] * ux.n[1], uysz = (long)uy.n[0] * uy.n[1], uzsz = (long)uz.n[0] * uz.n[1];
    const long fsz = (long)frc.n[0] * frc.n[1], dsz = (long)dv.n[0] * dv.n[1];
    auto qcol = q.gp() + q.off(ci, cj, k0 - 3) + q.cs * n;
    auto uxp = ux.gp() + ux.off(ci, cj, k0 - 1), uyp = uy.gp() + uy.off(ci, cj, k0 - 1), uzp = uz.gp() + uz.off(ci, cj, k0 - 1);
    auto frp = frc.gp() + frc.off(ci, cj, k0 - 1) + (has_force ? frc.cs * n : 0);
    auto dvp = dv.gp() + dv.off(ci, cj, k0 - 1);
    // state plane with 3 ghost cells: entries tid + r NT
    long qpo[NQ]; bool qpv[NQ];
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
        const int e = tid + r * NT, qi = tx0 - 3 + e % QW, qj = ty0 - 3 + e / QW;
        qpv[r] = e < QS && qi <= txe + 3 && qj <= tye + 3;
        qpo[r] = qpv[r] ? q.off(qi, qj, k0 - 1) + q.cs * n : 0;
    }
    // z-ring of the own column: r0..r4 = q(P-2..P+2) in iteration P
    double r0 = 0., r1v = 0., r2 = 0., r3 = 0., r4 = 0.;
    if (act) { r1v = qcol[0]; r2 = qcol[qsz]; r3 = qcol[2 * qsz]; r4 = qcol[3 * qsz]; }
    qcol += 4 * qsz;
    // prefetch registers for the first iteration
    double pq = 0., pmx = 0., pmy = 0., pmz = 0., pfr = 0., pdv = 0., pQ[NQ];
    if (act) {
        pq = qcol[0]; pmx = uxp[0]; pmy = uyp[0]; pmz = uzp[0];
        if (has_force) pfr = frp[0];
        if (has_divu) pdv = dvp[0];
    }
#pragma unroll
    for (int r = 0; r < NQ; ++r) pQ[r] = qpv[r] ? q.gp()[qpo[r]] : 0.;
    // values of plane P-1 kept from the previous iteration
    double xl1 = 0., xh1 = 0., yl1 = 0., yh1 = 0., qxm1 = 0., qym1 = 0., mx1 = 0., my1 = 0., mz1 = 0., fr1 = 0., dv1 = 0., slz1 = 0., Zprev = 0.;
    double smz1 = 0., spz1 = 0.;
    const double ax = dx1 * dx2, ay = dx2 * dx0, az = dx0 * dx1, qvol = 1.0 / (dx0 * dx1 * dx2);
    const double rdx0 = 1.0 / dx0, rdx1 = 1.0 / dx1, rdx2 = 1.0 / dx2;
    (void)rdx0; (void)rdx1; (void)rdx2;
    const bool store_edge = edge_t.t[0] != nullptr, store_flux = flux_t.t[0] != nullptr;

    int it = 0;
    for (int Pk = k0 - 1; Pk <= k1 + 1; ++Pk, ++it) {
        const int s = it & 1, sp = s ^ 1;
        // ---------------- stage 0
        r0 = r1v; r1v = r2; r2 = r3; r3 = r4; r4 = pq;
        const double mx0 = pmx, my0 = pmy, mz0 = pmz, fr0 = pfr, dv0 = pdv;
        if (act) { MX[s][o] = mx0; MY[s][o] = my0; MZ[s][o] = mz0; FR[s][o] = fr0; DV[s][o] = dv0; }
#pragma unroll
        for (int r = 0; r < NQ; ++r) { const int e = tid + r * NT; if (e < QS) Q[e] = pQ[r]; }
        if (Pk <= k1) {
            qcol += qsz; uxp += uxsz; uyp += uysz; uzp += uzsz; frp += fsz; dvp += dsz;
            if (act) {
                pq = qcol[0]; pmx = uxp[0]; pmy = uyp[0]; pmz = uzp[0];
                if (has_force) pfr = frp[0];
                if (has_divu) pdv = dvp[0];
            }
#pragma unroll
            for (int r = 0; r < NQ; ++r) { qpo[r] += qsz; pQ[r] = qpv[r] ? q.gp()[qpo[r]] : 0.; }
        }
        __syncthreads();
        // ---------------- stage A: pass 1
        double xl0, xh0, yl0, yh0, zl, zh, qxm0, qym0;
        {
            const double* qc = Q + qo;
            const double a3 = qc[-3], a2 = qc[-2], a1 = qc[-1], c0 = qc[0], b1 = qc[1], b2 = qc[2];
            qxm0 = a1;
            if constexpr (PPM) {
                xl0 = ppm_trace(qc - 1, 1L, edl0, edh0, ci - 1, dl0, dh0, mx0, dtdx0, true);
                xh0 = ppm_trace(qc, 1L, edl0, edh0, ci, dl0, dh0, mx0, dtdx0, false);
                (void)a3; (void)a2; (void)b1; (void)b2;
            } else {
            double sll, slh;
            if constexpr (IAMRX_GOD_ROW16 && !BCS && PW == 16) { (void)a3; slope4_row16(a2, a1, c0, b1, b2, li == 0, ci == txe + 1, sll, slh); }
            else {
            sll = slope4v(a3, a2, a1, c0, b1, edl0, edh0, ci - 1, dl0, dh0);
            slh = slope4v(a2, a1, c0, b1, b2, edl0, edh0, ci, dl0, dh0);
            }
            xh0 = c0 + 0.5 * (-1.0 - mx0 * dtdx0) * slh;
            xl0 = a1 + 0.5 * (1.0 - mx0 * dtdx0) * sll;
            }
            if (early_force) { xl0 += hdt * FR[s][oxm]; xh0 += hdt * fr0; }
            if (np0) trans_bc_v(a1, c0, ci, nv0, xl0, xh0, bl0, bh0, dl0, dh0);
            if (act) EX[s][o] = upwind_fu(mx0, xl0, xh0);
        }
        {
            const double* qc = Q + qo;
            const double a3 = qc[-3 * QW], a2 = qc[-2 * QW], a1 = qc[-QW], c0 = qc[0], b1 = qc[QW], b2 = qc[2 * QW];
            qym0 = a1;
            if constexpr (PPM) {
                yl0 = ppm_trace(qc - QW, (long)QW, edl1, edh1, cj - 1, dl1, dh1, my0, dtdx1, true);
                yh0 = ppm_trace(qc, (long)QW, edl1, edh1, cj, dl1, dh1, my0, dtdx1, false);
                (void)a3; (void)a2; (void)b1; (void)b2;
            } else {
            const double sll = slope4v(a3, a2, a1, c0, b1, edl1, edh1, cj - 1, dl1, dh1);
            const double slh = slope4v(a2, a1, c0, b1, b2, edl1, edh1, cj, dl1, dh1);
            yh0 = c0 + 0.5 * (-1.0 - my0 * dtdx1) * slh;
            yl0 = a1 + 0.5 * (1.0 - my0 * dtdx1) * sll;
            }
            if (early_force) { yl0 += hdt * FR[s][oym]; yh0 += hdt * fr0; }
            if (np1) trans_bc_v(a1, c0, cj, nv1, yl0, yh0, bl1, bh1, dl1, dh1);
            if (act) EY[s][o] = upwind_fu(my0, yl0, yh0);
        }
        double slz0 = 0.0, smz0 = 0.0, spz0 = 0.0;
        {
            if constexpr (PPM) {
                // the parabola of cell P from the z-ring; the one of cell P-1 is kept from the iteration before (both faces of a cell
                // are traced with the velocity of the face in question)
                const double zr[5] = {r0, r1v, r2, r3, r4};
                ppm_edges(zr + 2, 1L, edl2, edh2, Pk, dl2, dh2, smz0, spz0);
                zh = ppm_state(r2, smz0, spz0, mz0, dtdx2, false);
                zl = ppm_state(r1v, smz1, spz1, mz0, dtdx2, true);
            } else {
            slz0 = slope4v(r0, r1v, r2, r3, r4, edl2, edh2, Pk, dl2, dh2);
            zh = r2 + 0.5 * (-1.0 - mz0 * dtdx2) * slz0;
            zl = r1v + 0.5 * (1.0 - mz0 * dtdx2) * slz1;
            }
            if (early_force) { zl += hdt * fr1; zh += hdt * fr0; }
            if (np2) trans_bc_v(r1v, r2, Pk, nv2, zl, zh, bl2, bh2, dl2, dh2);
            if (act) EZ[s][o] = upwind_fu(mz0, zl, zh);
        }
        __syncthreads();
        // ---------------- stage B: corner coupling
        {
            const double myA = MY[s][oxm], myB = MY[s][oxm_yp], myD = MY[s][oyp];
            const double mxA = MX[s][oym], mxB = MX[s][oxp_ym], mxD = MX[s][oxp];
            const double mxp1 = MX[sp][oxp], myp1 = MY[sp][oyp];
            const double mzxm1 = MZ[sp][oxm], mzxm0 = MZ[s][oxm], mzym1 = MZ[sp][oym], mzym0 = MZ[s][oym];
            const double dvxm0 = has_divu ? DV[s][oxm] : 0., dvym0 = has_divu ? DV[s][oym] : 0.;
            const double dvxm1 = has_divu ? DV[sp][oxm] : 0., dvym1 = has_divu ? DV[sp][oym] : 0.;
            // x-face corrected by y, plane P
            const double cxy = corner_core(xl0, xh0, qxm0, r2, mx0, myA, myB, my0, myD, EY[s][oxm], EY[s][oxm_yp], EY[s][o], EY[s][oyp],
                                           dvxm0, dv0, conserv, co1, dt3, dx1, np0, nv0, ci, bl0, bh0, dl0, dh0);
            // y-face corrected by x, plane P
            const double cyx = corner_core(yl0, yh0, qym0, r2, my0, mxA, mxB, mx0, mxD, EX[s][oym], EX[s][oxp_ym], EX[s][o], EX[s][oxp],
                                           dvym0, dv0, conserv, co0, dt3, dx0, np1, nv1, cj, bl1, bh1, dl1, dh1);
            // z-face P corrected by x / by y: low-side cell = plane P-1
            const double czx = corner_core(zl, zh, r1v, r2, mz0, mx1, mxp1, mx0, mxD, EX[sp][o], EX[sp][oxp], EX[s][o], EX[s][oxp],
                                           dv1, dv0, conserv, co0, dt3, dx0, np2, nv2, Pk, bl2, bh2, dl2, dh2);
            const double czy = corner_core(zl, zh, r1v, r2, mz0, my1, myp1, my0, myD, EY[sp][o], EY[sp][oyp], EY[s][o], EY[s][oyp],
                                           dv1, dv0, conserv, co1, dt3, dx1, np2, nv2, Pk, bl2, bh2, dl2, dh2);
            // x- / y-face of plane P-1 corrected by z
            const double cxz = corner_core(xl1, xh1, qxm1, r1v, mx1, mzxm1, mzxm0, mz1, mz0, EZ[sp][oxm], EZ[s][oxm], EZ[sp][o], EZ[s][o],
                                           dvxm1, dv1, conserv, co2, dt3, dx2, np0, nv0, ci, bl0, bh0, dl0, dh0);
            const double cyz = corner_core(yl1, yh1, qym1, r1v, my1, mzym1, mzym0, mz1, mz0, EZ[sp][oym], EZ[s][oym], EZ[sp][o], EZ[s][o],
                                           dvym1, dv1, conserv, co2, dt3, dx2, np1, nv1, cj, bl1, bh1, dl1, dh1);
            if (act) { CXY[s][o] = cxy; CYX[s][o] = cyx; CZX[s][o] = czx; CZY[s][o] = czy; CXZ[o] = cxz; CYZ[o] = cyz; }
        }
        __syncthreads();
        // ---------------- stage C: final edge states: x, y of plane P-1; z of face P
        double Xe, Ye, Ze, uxh, uyh;
        {
            const double mxp1 = MX[sp][oxp], myp1 = MY[sp][oyp], mxp0 = MX[s][oxp], myp0 = MY[s][oyp];
            uxh = mxp1; uyh = myp1;
            const double mzxm1 = MZ[sp][oxm], mzxm0 = MZ[s][oxm], mzym1 = MZ[sp][oym], mzym0 = MZ[s][oym];
            Xe = final_core<false>(xl1, xh1, mx1, MY[sp][oxm], MY[sp][oxm_yp], my1, myp1, mzxm1, mzxm0, mz1, mz0,
                            CYZ[oxm], CYZ[oxm_yp], CYZ[o], CYZ[oyp], CZY[sp][oxm], CZY[s][oxm], CZY[sp][o], CZY[s][o],
                            qxm1, r1v, late_force ? FR[sp][oxm] : 0., fr1, has_divu ? DV[sp][oxm] : 0., dv1,
                            conserv, has_divu, late_force, dt, dx1, dx2, np0, nv0, ci, bl0, bh0, dl0, dh0);
            Ye = final_core<false>(yl1, yh1, my1, MX[sp][oym], MX[sp][oxp_ym], mx1, mxp1, mzym1, mzym0, mz1, mz0,
                            CXZ[oym], CXZ[oxp_ym], CXZ[o], CXZ[oxp], CZX[sp][oym], CZX[s][oym], CZX[sp][o], CZX[s][o],
                            qym1, r1v, late_force ? FR[sp][oym] : 0., fr1, has_divu ? DV[sp][oym] : 0., dv1,
                            conserv, has_divu, late_force, dt, dx0, dx2, np1, nv1, cj, bl1, bh1, dl1, dh1);
            Ze = final_core<false>(zl, zh, mz0, mx1, mxp1, mx0, mxp0, my1, myp1, my0, myp0,
                            CXY[sp][o], CXY[sp][oxp], CXY[s][o], CXY[s][oxp], CYX[sp][o], CYX[sp][oyp], CYX[s][o], CYX[s][oyp],
                            r1v, r2, fr1, fr0, dv1, dv0,
                            conserv, has_divu, late_force, dt, dx0, dx1, np2, nv2, Pk, bl2, bh2, dl2, dh2);
            // x / y states for the neighbour threads: the E_z / E_x slots of face / plane P-1 are free from here on
            if (act) { EZ[sp][o] = Xe; EX[sp][o] = Ye; }
        }
        __syncthreads();
        // ---------------- stage D: plane P-1
        {
            const int k = Pk - 1;
            const double exh = EZ[sp][oxp], eyh = EX[sp][oyp];
            if (act && k >= k0 && k <= k1) {
                const double fxl = Xe * mx1 * ax, fyl = Ye * my1 * ay, fzl = Zprev * mz1 * az;
                if (in_tile) {
                    const double fxh = exh * uxh * ax, fyh = eyh * uyh * ay, fzh = Ze * mz0 * az;
                    const double divum = 1.0 * (IAMRX_DIVDX(uxh - mx1, dx0, rdx0) + IAMRX_DIVDX(uyh - my1, dx1, rdx1) + IAMRX_DIVDX(mz0 - mz1, dx2, rdx2));
                    double upd = -1.0 * qvol * ((fxh - fxl) + (fyh - fyl) + (fzh - fzl));
                    if (!conserv) {
                        double qavg = Xe + exh + Ye + eyh + Zprev + Ze;
                        qavg *= 1.0 / 6.0;
                        upd += qavg * divum;
                    }
                    aofs(ci, cj, k, acomp + n) = -upd;
                }
                // faces: every face is written by the tile on its high side, the faces on the high end of the box by the last tile
                const bool wx = cj >= ty0 && cj <= tye && ci >= tx0 && (ci <= txe || ci == b.hi[0] + 1);
                const bool wy = ci >= tx0 && ci <= txe && cj >= ty0 && (cj <= tye || cj == b.hi[1] + 1);
                if (store_edge) {
                    if (wx) edge_t.t[0][fab](ci, cj, k, n) = Xe;
                    if (wy) edge_t.t[1][fab](ci, cj, k, n) = Ye;
                    if (in_tile) edge_t.t[2][fab](ci, cj, k, n) = Zprev;
                }
                if (store_flux) {
                    if (wx) flux_t.t[0][fab](ci, cj, k, n) = fxl;
                    if (wy) flux_t.t[1][fab](ci, cj, k, n) = fyl;
                    if (in_tile) flux_t.t[2][fab](ci, cj, k, n) = fzl;
                }
            }
            if (in_tile && k == b.hi[2]) {
                // the z-face on the high end of the box
                if (store_edge) edge_t.t[2][fab](ci, cj, k + 1, n) = Ze;
                if (store_flux) flux_t.t[2][fab](ci, cj, k + 1, n) = Ze * mz0 * az;
            }
        }
        xl1 = xl0; xh1 = xh0; yl1 = yl0; yh1 = yh0; qxm1 = qxm0; qym1 = qym0;
        mx1 = mx0; my1 = my0; mz1 = mz0; fr1 = fr0; dv1 = dv0; slz1 = slz0; Zprev = Ze;
        smz1 = smz0; spz1 = spz0;
    }
}

// IAMRX_GODUNOV_Z=0 selects the multi-pass kernels (read at every call: tests/test_gpu_godunov_fused.py compares the two paths)
static bool use_z_kernel()
{
    return tune("GODUNOV_Z", 1) != 0;
}

template <int TX, int TY, int WPE, bool BCS, bool PPM>
static void launch_god_z(const Layout& l, MultiFab& aofs, int acomp, const MultiFab& S, int ncomp, const MultiFab* force, const MultiFab* divu,
                         MultiFab* const umac[3], MultiFab* const edge_out[3], MultiFab* const flux_out[3], const GodParams* dP, int sel = 0,
                         const Geometry* sg = nullptr)
{
    constexpr int NT = (((TX + 2) * (TY + 2)) + 63) / 64 * 64;
    const int ntx = (l.max_len[0] + TX - 1) / TX, nty = (l.max_len[1] + TY - 1) / TY;
    const int kc_env = (int)tune("GODUNOV_ZKC", 0);
    const int kc = kc_env > 0 ? kc_env : std::min(64, std::max(8, l.max_len[2] / 4));       // planes marched per workgroup
    const GodChunks zc = god_chunks(l.max_len[2], kc, sel != 0 && !sg->periodic[2]);
    int total = ntx * nty * zc.n, ntiles = 0, lx = 0;
    unsigned gy = (unsigned)l.nlocal();
    const int4* tiles = nullptr;
    if (sel != 0) {           // this launch's share of the tiles of a domain with walls (god_tile_lists)
        const GodTileList tl = god_tile_lists(l, *sg, TX, TY, zc);
        tiles = tl.d[sel - 1]; ntiles = tl.n[sel - 1];
        if (ntiles == 0) return;
        total = ntiles; gy = 1u; lx = tl.xcd_cnt[sel - 1];
    }
    const int xcd_cnt = sel != 0 ? lx : (total >= 64 ? (total + 7) / 8 : 0);
    dim3 grid((unsigned)(xcd_cnt > 0 ? 8 * xcd_cnt : total), gy, (unsigned)ncomp);
    GodTabs3 et{{nullptr, nullptr, nullptr}}, ftb{{nullptr, nullptr, nullptr}};
    if (edge_out && edge_out[0]) for (int d = 0; d < 3; ++d) et.t[d] = edge_out[d]->d_tab;
    if (flux_out && flux_out[0]) for (int d = 0; d < 3; ++d) ftb.t[d] = flux_out[d]->d_tab;
    const bool rec = kernel_probe_begin(PROBE_GOD_Z, (long)l.max_len[0] * l.max_len[1] * l.max_len[2]);
    hipLaunchKernelGGL((k_god_z<TX, TY, NT, WPE, BCS, PPM>), grid, dim3(NT), 0, Context::get().stream, l.d_boxes, S.d_tab, force ? force->d_tab : nullptr,
                       divu ? divu->d_tab : nullptr, umac[0]->d_tab, umac[1]->d_tab, umac[2]->d_tab, aofs.d_tab, acomp, et, ftb, dP,
                       ntx, nty, zc, tiles, ntiles, xcd_cnt);
    if (rec) kernel_probe_end(PROBE_GOD_Z);
}

// -------------------------------------------------------------------------------- fused z-marching velocity prediction (PLM)
// k_pred_z: ExtrapVelToFaces in one launch, organised as k_god_z.  The traced states use the cell-centred velocity of the trace
// direction, the advective velocities ad_x / ad_y / ad_z (upwinded normal components) take the place of the mac velocities and live
// in LDS like the pass-1 states; the final x- / y- / z-face state is that of u / v / w, so the corner-coupled state C_{T|O} is formed
// for the component 3-T-O only: E_x(v,w), E_y(u,w), E_z(u,v), six corner arrays, three final states per column and plane.
// Three barriers per plane (the final states go straight to umac).
template <int TX, int TY, int NT, bool BCS, bool PPM>
__global__ void __launch_bounds__(NT, 2) k_pred_z(const BoxD* __restrict__ boxes, const FabD* __restrict__ qt, const FabD* __restrict__ ft,
    const FabD* __restrict__ uxt, const FabD* __restrict__ uyt, const FabD* __restrict__ uzt, const GodParams* __restrict__ Pp,
    int ntx, int nty, GodChunks zc, const int4* __restrict__ tiles, int ntiles, int xcd_cnt)
{
    constexpr int PW = TX + 2, PH = TY + 2, PS = PW * PH, QW = TX + 6, QH = TY + 6, QS = QW * QH;
    constexpr int NQ = (QS + NT - 1) / NT;
    static_assert(NT >= PS, "one thread per column of the grown tile");
    __shared__ double Qb[3][QS + QW];
    __shared__ double AX[2][PS], AY[2][PS], AZ[2][PS], FR[2][3][PS];
    __shared__ double EXv[2][PS], EXw[2][PS], EYu[2][PS], EYw[2][PS], EZu[2][PS], EZv[2][PS];
    __shared__ double CZX[2][PS], CZY[2][PS], CXY[2][PS], CYX[2][PS], CXZ[PS], CYZ[PS];
    const GodParams& P = *Pp;
    int bid = blockIdx.x;
    if (xcd_cnt > 0) {
        bid = (bid & 7) * xcd_cnt + (bid >> 3);          // XCD-aware order, see make_tiling
        if (bid >= (tiles ? ntiles : ntx * nty * zc.n)) return;
    }
    // the tile: from the launch's list (god_tile_lists) or from the full tile grid of the box blockIdx.y
    int fab = blockIdx.y, tix, tiy, kci;
    if (tiles) { const int4 t = tiles[bid]; if (t.x < 0) return; fab = t.x; tix = t.y; tiy = t.z; kci = t.w; }
    else { tix = bid % ntx; const int r1 = bid / ntx; tiy = r1 % nty; kci = r1 / nty; }
    const BoxD b = boxes[fab];
    const int tx0 = b.lo[0] + tix * TX, ty0 = b.lo[1] + tiy * TY, k0 = b.lo[2] + zc.start[kci];
    if (tx0 > b.hi[0] || ty0 > b.hi[1] || k0 > b.hi[2]) return;
    const int txe = min(tx0 + TX - 1, b.hi[0]), tye = min(ty0 + TY - 1, b.hi[1]), k1 = min(b.lo[2] + zc.start[kci + 1] - 1, b.hi[2]);
    const int tid = threadIdx.x;
    const int li = tid % PW, lj = tid / PW;
    const bool act = tid < PS && tx0 - 1 + li <= txe + 1 && ty0 - 1 + lj <= tye + 1;
    const int ci = act ? tx0 - 1 + li : tx0, cj = act ? ty0 - 1 + lj : ty0;
    const int o = (ci - (tx0 - 1)) + PW * (cj - (ty0 - 1));
    const int oxm = ci > tx0 - 1 ? o - 1 : o, oxp = ci < txe + 1 ? o + 1 : o;
    const int oym = cj > ty0 - 1 ? o - PW : o, oyp = cj < tye + 1 ? o + PW : o;
    const int oxm_yp = oxm + (oyp - o), oxp_ym = oxp + (oym - o);
    const int qo = QW + (ci - (tx0 - 3)) + QW * (cj - (ty0 - 3));
    const bool in_tile = act && ci >= tx0 && ci <= txe && cj >= ty0 && cj <= tye;
    const bool wx = act && cj >= ty0 && cj <= tye && ci >= tx0 && (ci <= txe || ci == b.hi[0] + 1);
    const bool wy = act && ci >= tx0 && ci <= txe && cj >= ty0 && (cj <= tye || cj == b.hi[1] + 1);

    const bool has_force = P.has_force != 0, fit = P.fit != 0;
    const bool early_force = fit && has_force, late_force = !fit && has_force;
    const double dt = P.dt, hdt = 0.5 * dt, dt3 = dt / 3.0;
    const double dx0 = P.dx[0], dx1 = P.dx[1], dx2 = P.dx[2];
    const double dtdx0 = dt / dx0, dtdx1 = dt / dx1, dtdx2 = dt / dx2;
    const double co0 = dt / (6.0 * dx0), co1 = dt / (6.0 * dx1), co2 = dt / (6.0 * dx2);
    const bool np0 = BCS && !P.bc.per[0], np1 = BCS && !P.bc.per[1], np2 = BCS && !P.bc.per[2];   // BCS = false: periodic in every direction
    const int dl0 = P.bc.dlo[0], dh0 = P.bc.dhi[0], dl1 = P.bc.dlo[1], dh1 = P.bc.dhi[1], dl2 = P.bc.dlo[2], dh2 = P.bc.dhi[2];

    const FabD q = qt[fab], ux = uxt[fab], uy = uyt[fab], uz = uzt[fab];
    FabD frc = q;
    if (has_force) frc = ft[fab];
    const long qsz = (long)q.n[0] * q.n[1], fsz = (long)frc.n[0] * frc.n[1];
    auto qcol = q.gp() + q.off(ci, cj, k0 - 3);
    auto frp = frc.gp() + frc.off(ci, cj, k0 - 1);
    long qpo[NQ]; bool qpv[NQ];
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
        const int e = tid + r * NT, qi = tx0 - 3 + e % QW, qj = ty0 - 3 + e / QW;
        qpv[r] = e < QS && qi <= txe + 3 && qj <= tye + 3;
        qpo[r] = qpv[r] ? q.off(qi, qj, k0 - 1) : 0;
    }
    // z-rings of the own column: rz[c][0..4] = q_c(P-2..P+2) in iteration P
    double rz[3][5];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rz[c][0] = 0.;
#pragma unroll
        for (int m = 0; m < 4; ++m) rz[c][m + 1] = act ? qcol[m * qsz + q.cs * c] : 0.;
    }
    qcol += 4 * qsz;
    double pq[3], pfr[3], pQ[3][NQ];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pq[c] = act ? qcol[q.cs * c] : 0.;
        pfr[c] = (act && has_force) ? frp[frc.cs * c] : 0.;
#pragma unroll
        for (int r = 0; r < NQ; ++r) pQ[c][r] = qpv[r] ? q.gp()[qpo[r] + q.cs * c] : 0.;
    }
    double xl1u = 0., xh1u = 0., xl1v = 0., xh1v = 0., yl1u = 0., yh1u = 0., yl1v = 0., yh1v = 0.;
    double qxm1u = 0., qxm1v = 0., qym1u = 0., qym1v = 0., fr1[3] = {0., 0., 0.}, slz1[3] = {0., 0., 0.};
    double smz1[3] = {0., 0., 0.}, spz1[3] = {0., 0., 0.};

    int it = 0;
    for (int Pk = k0 - 1; Pk <= k1 + 1; ++Pk, ++it) {
        const int s = it & 1, sp = s ^ 1;
        // ---------------- stage 0
        double fr0[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rz[c][0] = rz[c][1]; rz[c][1] = rz[c][2]; rz[c][2] = rz[c][3]; rz[c][3] = rz[c][4]; rz[c][4] = pq[c];
            fr0[c] = pfr[c];
            if (act && has_force) FR[s][c][o] = fr0[c];
#pragma unroll
            for (int r = 0; r < NQ; ++r) { const int e = tid + r * NT; if (e < QS) Qb[c][QW + e] = pQ[c][r]; }
        }
        if (Pk <= k1) {
            qcol += qsz; frp += fsz;
#pragma unroll
            for (int r = 0; r < NQ; ++r) qpo[r] += qsz;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (act) { pq[c] = qcol[q.cs * c]; if (has_force) pfr[c] = frp[frc.cs * c]; }
#pragma unroll
                for (int r = 0; r < NQ; ++r) pQ[c][r] = qpv[r] ? q.gp()[qpo[r] + q.cs * c] : 0.;
            }
        }
        __syncthreads();
        // ---------------- stage A: pass 1 (traces with the cell-centred velocity of the trace direction)
        double xl0u, xh0u, xl0v, xh0v, xl0w, xh0w, yl0u, yh0u, yl0v, yh0v, yl0w, yh0w, zlu, zhu, zlv, zhv, zlw, zhw;
        double qxm0u, qxm0v, qxm0w, qym0u, qym0v, qym0w, adx, ady, adz, slz0[3];
        double smz0[3] = {0., 0., 0.}, spz0[3] = {0., 0., 0.};
        {
            const double vlo = Qb[0][qo - 1], vhi = Qb[0][qo];
            double l[3], h[3], am[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double* qc = &Qb[c][qo];
                const double a3 = qc[-3], a2 = qc[-2], a1 = qc[-1], c0 = qc[0], b1 = qc[1], b2 = qc[2];
                const int bl = P.bc.bc[c].lo[0], bh = P.bc.bc[c].hi[0];
                const bool edlo = np0 && ed_or_ho(bl), edhi = np0 && ed_or_ho(bh);
                if constexpr (PPM) {
                    l[c] = ppm_trace(qc - 1, 1L, edlo, edhi, ci - 1, dl0, dh0, vlo, dtdx0, true);
                    h[c] = ppm_trace(qc, 1L, edlo, edhi, ci, dl0, dh0, vhi, dtdx0, false);
                    (void)a3; (void)a2; (void)b1; (void)b2;
                } else {
                double sll, slh;
                if constexpr (IAMRX_GOD_ROW16 && !BCS && PW == 16) { (void)a3; slope4_row16(a2, a1, c0, b1, b2, li == 0, ci == txe + 1, sll, slh); }
                else {
                sll = slope4v(a3, a2, a1, c0, b1, edlo, edhi, ci - 1, dl0, dh0);
                slh = slope4v(a2, a1, c0, b1, b2, edlo, edhi, ci, dl0, dh0);
                }
                h[c] = c0 + 0.5 * (-1.0 - vhi * dtdx0) * slh;
                l[c] = a1 + 0.5 * (1.0 - vlo * dtdx0) * sll;
                }
                if (early_force) { l[c] += hdt * FR[s][c][oxm]; h[c] += hdt * fr0[c]; }
                if (np0) trans_bc_v(a1, c0, ci, c == 0, l[c], h[c], bl, bh, dl0, dh0);
                am[c] = a1;
            }
            const double st = ((l[0] + h[0]) >= 0.) ? l[0] : h[0];
            const bool ltm = ((l[0] <= 0. && h[0] >= 0.) || (fabs(l[0] + h[0]) < SMALL_VEL));
            adx = ltm ? 0. : st;
            xl0u = l[0]; xh0u = h[0]; xl0v = l[1]; xh0v = h[1]; xl0w = l[2]; xh0w = h[2];
            qxm0u = am[0]; qxm0v = am[1]; qxm0w = am[2];
            if (act) { AX[s][o] = adx; EXv[s][o] = upwind_fu(adx, l[1], h[1]); EXw[s][o] = upwind_fu(adx, l[2], h[2]); }
        }
        {
            const double vlo = Qb[1][qo - QW], vhi = Qb[1][qo];
            double l[3], h[3], am[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double* qc = &Qb[c][qo];
                const double a3 = qc[-3 * QW], a2 = qc[-2 * QW], a1 = qc[-QW], c0 = qc[0], b1 = qc[QW], b2 = qc[2 * QW];
                const int bl = P.bc.bc[c].lo[1], bh = P.bc.bc[c].hi[1];
                const bool edlo = np1 && ed_or_ho(bl), edhi = np1 && ed_or_ho(bh);
                if constexpr (PPM) {
                    l[c] = ppm_trace(qc - QW, (long)QW, edlo, edhi, cj - 1, dl1, dh1, vlo, dtdx1, true);
                    h[c] = ppm_trace(qc, (long)QW, edlo, edhi, cj, dl1, dh1, vhi, dtdx1, false);
                    (void)a3; (void)a2; (void)b1; (void)b2;
                } else {
                const double sll = slope4v(a3, a2, a1, c0, b1, edlo, edhi, cj - 1, dl1, dh1);
                const double slh = slope4v(a2, a1, c0, b1, b2, edlo, edhi, cj, dl1, dh1);
                h[c] = c0 + 0.5 * (-1.0 - vhi * dtdx1) * slh;
                l[c] = a1 + 0.5 * (1.0 - vlo * dtdx1) * sll;
                }
                if (early_force) { l[c] += hdt * FR[s][c][oym]; h[c] += hdt * fr0[c]; }
                if (np1) trans_bc_v(a1, c0, cj, c == 1, l[c], h[c], bl, bh, dl1, dh1);
                am[c] = a1;
            }
            const double st = ((l[1] + h[1]) >= 0.) ? l[1] : h[1];
            const bool ltm = ((l[1] <= 0. && h[1] >= 0.) || (fabs(l[1] + h[1]) < SMALL_VEL));
            ady = ltm ? 0. : st;
            yl0u = l[0]; yh0u = h[0]; yl0v = l[1]; yh0v = h[1]; yl0w = l[2]; yh0w = h[2];
            qym0u = am[0]; qym0v = am[1]; qym0w = am[2];
            if (act) { AY[s][o] = ady; EYu[s][o] = upwind_fu(ady, l[0], h[0]); EYw[s][o] = upwind_fu(ady, l[2], h[2]); }
        }
        {
            const double vlo = rz[2][1], vhi = rz[2][2];
            double l[3], h[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int bl = P.bc.bc[c].lo[2], bh = P.bc.bc[c].hi[2];
                const bool edlo = np2 && ed_or_ho(bl), edhi = np2 && ed_or_ho(bh);
                if constexpr (PPM) {
                    slz0[c] = 0.0;
                    ppm_edges(&rz[c][2], 1L, edlo, edhi, Pk, dl2, dh2, smz0[c], spz0[c]);
                    h[c] = ppm_state(rz[c][2], smz0[c], spz0[c], vhi, dtdx2, false);
                    l[c] = ppm_state(rz[c][1], smz1[c], spz1[c], vlo, dtdx2, true);
                } else {
                slz0[c] = slope4v(rz[c][0], rz[c][1], rz[c][2], rz[c][3], rz[c][4], edlo, edhi, Pk, dl2, dh2);
                h[c] = rz[c][2] + 0.5 * (-1.0 - vhi * dtdx2) * slz0[c];
                l[c] = rz[c][1] + 0.5 * (1.0 - vlo * dtdx2) * slz1[c];
                }
                if (early_force) { l[c] += hdt * fr1[c]; h[c] += hdt * fr0[c]; }
                if (np2) trans_bc_v(rz[c][1], rz[c][2], Pk, c == 2, l[c], h[c], bl, bh, dl2, dh2);
            }
            const double st = ((l[2] + h[2]) >= 0.) ? l[2] : h[2];
            const bool ltm = ((l[2] <= 0. && h[2] >= 0.) || (fabs(l[2] + h[2]) < SMALL_VEL));
            adz = ltm ? 0. : st;
            zlu = l[0]; zhu = h[0]; zlv = l[1]; zhv = h[1]; zlw = l[2]; zhw = h[2];
            if (act) { AZ[s][o] = adz; EZu[s][o] = upwind_fu(adz, l[0], h[0]); EZv[s][o] = upwind_fu(adz, l[1], h[1]); }
        }
        __syncthreads();
        // ---------------- stage B: corner coupling (convective form)
        double frxm1 = 0., frym1 = 0.;
        {
            const double ayA = AY[s][oxm], ayB = AY[s][oxm_yp], ayD = AY[s][oyp];
            const double axA = AX[s][oym], axB = AX[s][oxp_ym], axD = AX[s][oxp];
            const double ax1 = AX[sp][o], ay1 = AY[sp][o], az1 = AZ[sp][o], axp1 = AX[sp][oxp], ayp1 = AY[sp][oyp];
            const double azxm1 = AZ[sp][oxm], azxm0 = AZ[s][oxm], azym1 = AZ[sp][oym], azym0 = AZ[s][oym];
            const int bxl_w = P.bc.bc[2].lo[0], bxh_w = P.bc.bc[2].hi[0], byl_w = P.bc.bc[2].lo[1], byh_w = P.bc.bc[2].hi[1];
            const int bzl_v = P.bc.bc[1].lo[2], bzh_v = P.bc.bc[1].hi[2], bzl_u = P.bc.bc[0].lo[2], bzh_u = P.bc.bc[0].hi[2];
            const int bxl_v = P.bc.bc[1].lo[0], bxh_v = P.bc.bc[1].hi[0], byl_u = P.bc.bc[0].lo[1], byh_u = P.bc.bc[0].hi[1];
            const double cxy = corner_core(xl0w, xh0w, qxm0w, rz[2][2], adx, ayA, ayB, ady, ayD, EYw[s][oxm], EYw[s][oxm_yp], EYw[s][o], EYw[s][oyp],
                                           0., 0., false, co1, dt3, dx1, np0, false, ci, bxl_w, bxh_w, dl0, dh0);
            const double cyx = corner_core(yl0w, yh0w, qym0w, rz[2][2], ady, axA, axB, adx, axD, EXw[s][oym], EXw[s][oxp_ym], EXw[s][o], EXw[s][oxp],
                                           0., 0., false, co0, dt3, dx0, np1, false, cj, byl_w, byh_w, dl1, dh1);
            const double czx = corner_core(zlv, zhv, rz[1][1], rz[1][2], adz, ax1, axp1, adx, axD, EXv[sp][o], EXv[sp][oxp], EXv[s][o], EXv[s][oxp],
                                           0., 0., false, co0, dt3, dx0, np2, false, Pk, bzl_v, bzh_v, dl2, dh2);
            const double czy = corner_core(zlu, zhu, rz[0][1], rz[0][2], adz, ay1, ayp1, ady, ayD, EYu[sp][o], EYu[sp][oyp], EYu[s][o], EYu[s][oyp],
                                           0., 0., false, co1, dt3, dx1, np2, false, Pk, bzl_u, bzh_u, dl2, dh2);
            const double cxz = corner_core(xl1v, xh1v, qxm1v, rz[1][1], ax1, azxm1, azxm0, az1, adz, EZv[sp][oxm], EZv[s][oxm], EZv[sp][o], EZv[s][o],
                                           0., 0., false, co2, dt3, dx2, np0, false, ci, bxl_v, bxh_v, dl0, dh0);
            const double cyz = corner_core(yl1u, yh1u, qym1u, rz[0][1], ay1, azym1, azym0, az1, adz, EZu[sp][oym], EZu[s][oym], EZu[sp][o], EZu[s][o],
                                           0., 0., false, co2, dt3, dx2, np1, false, cj, byl_u, byh_u, dl1, dh1);
            if (act) { CXY[s][o] = cxy; CYX[s][o] = cyx; CZX[s][o] = czx; CZY[s][o] = czy; CXZ[o] = cxz; CYZ[o] = cyz; }
            if (late_force) { frxm1 = FR[sp][0][oxm]; frym1 = FR[sp][1][oym]; }
        }
        __syncthreads();
        // ---------------- stage C: u on the x-faces and v on the y-faces of plane P-1, w on the z-face P
        {
            const double ax1 = AX[sp][o], ay1 = AY[sp][o], az1 = AZ[sp][o];
            const double axp1 = AX[sp][oxp], ayp1 = AY[sp][oyp], axp0 = AX[s][oxp], ayp0 = AY[s][oyp];
            const double azxm1 = AZ[sp][oxm], azxm0 = AZ[s][oxm], azym1 = AZ[sp][oym], azym0 = AZ[s][oym];
            const double Xe = final_core<true>(xl1u, xh1u, ax1, AY[sp][oxm], AY[sp][oxm_yp], ay1, ayp1, azxm1, azxm0, az1, adz,
                            CYZ[oxm], CYZ[oxm_yp], CYZ[o], CYZ[oyp], CZY[sp][oxm], CZY[s][oxm], CZY[sp][o], CZY[s][o],
                            qxm1u, rz[0][1], frxm1, fr1[0], 0., 0., false, false, late_force, dt, dx1, dx2,
                            np0, true, ci, P.bc.bc[0].lo[0], P.bc.bc[0].hi[0], dl0, dh0);
            const double Ye = final_core<true>(yl1v, yh1v, ay1, AX[sp][oym], AX[sp][oxp_ym], ax1, axp1, azym1, azym0, az1, adz,
                            CXZ[oym], CXZ[oxp_ym], CXZ[o], CXZ[oxp], CZX[sp][oym], CZX[s][oym], CZX[sp][o], CZX[s][o],
                            qym1v, rz[1][1], frym1, fr1[1], 0., 0., false, false, late_force, dt, dx0, dx2,
                            np1, true, cj, P.bc.bc[1].lo[1], P.bc.bc[1].hi[1], dl1, dh1);
            const double Ze = final_core<true>(zlw, zhw, adz, ax1, axp1, adx, axp0, ay1, ayp1, ady, ayp0,
                            CXY[sp][o], CXY[sp][oxp], CXY[s][o], CXY[s][oxp], CYX[sp][o], CYX[sp][oyp], CYX[s][o], CYX[s][oyp],
                            rz[2][1], rz[2][2], fr1[2], fr0[2], 0., 0., false, false, late_force, dt, dx0, dx1,
                            np2, true, Pk, P.bc.bc[2].lo[2], P.bc.bc[2].hi[2], dl2, dh2);
            const int k = Pk - 1;
            if (k >= k0 && k <= k1) {
                if (wx) ux(ci, cj, k, 0) = Xe;
                if (wy) uy(ci, cj, k, 0) = Ye;
            }
            if (in_tile && Pk >= k0 && (Pk <= k1 || Pk == b.hi[2] + 1)) uz(ci, cj, Pk, 0) = Ze;
        }
        xl1u = xl0u; xh1u = xh0u; xl1v = xl0v; xh1v = xh0v; yl1u = yl0u; yh1u = yh0u; yl1v = yl0v; yh1v = yh0v;
        qxm1u = qxm0u; qxm1v = qxm0v; qym1u = qym0u; qym1v = qym0v;
#pragma unroll
        for (int c = 0; c < 3; ++c) { fr1[c] = fr0[c]; slz1[c] = slz0[c]; smz1[c] = smz0[c]; spz1[c] = spz0[c]; }
    }
}

template <int TX, int TY, bool BCS, bool PPM>
static void launch_pred_z(const Layout& l, const MultiFab& vel, const MultiFab* force, MultiFab* const umac[3], const GodParams* dP, int sel = 0,
                          const Geometry* sg = nullptr)
{
    constexpr int NT = (((TX + 2) * (TY + 2)) + 63) / 64 * 64;
    const int ntx = (l.max_len[0] + TX - 1) / TX, nty = (l.max_len[1] + TY - 1) / TY;
    const int kc_env = (int)tune("GODUNOV_ZKC", 0);
    const int kc = kc_env > 0 ? kc_env : std::min(64, std::max(8, l.max_len[2] / 4));
    const GodChunks zc = god_chunks(l.max_len[2], kc, sel != 0 && !sg->periodic[2]);
    int total = ntx * nty * zc.n, ntiles = 0, lx = 0;
    unsigned gy = (unsigned)l.nlocal();
    const int4* tiles = nullptr;
    if (sel != 0) {           // this launch's share of the tiles of a domain with walls (god_tile_lists)
        const GodTileList tl = god_tile_lists(l, *sg, TX, TY, zc);
        tiles = tl.d[sel - 1]; ntiles = tl.n[sel - 1];
        if (ntiles == 0) return;
        total = ntiles; gy = 1u; lx = tl.xcd_cnt[sel - 1];
    }
    const int xcd_cnt = sel != 0 ? lx : (total >= 64 ? (total + 7) / 8 : 0);
    dim3 grid((unsigned)(xcd_cnt > 0 ? 8 * xcd_cnt : total), gy, 1u);
    const bool rec = kernel_probe_begin(PROBE_PRED_Z, (long)l.max_len[0] * l.max_len[1] * l.max_len[2]);
    hipLaunchKernelGGL((k_pred_z<TX, TY, NT, BCS, PPM>), grid, dim3(NT), 0, Context::get().stream, l.d_boxes, vel.d_tab, force ? force->d_tab : nullptr,
                       umac[0]->d_tab, umac[1]->d_tab, umac[2]->d_tab, dP, ntx, nty, zc, tiles, ntiles, xcd_cnt);
    if (rec) kernel_probe_end(PROBE_PRED_Z);
}

static void godunov_pred_z(const Layout& l, const MultiFab& vel, const MultiFab* force, MultiFab* const umac[3], const GodParams* dP, bool bcs, bool ppm, const Geometry& g)
{
    // 14 x 14 cells per workgroup (grown tile 16 x 16 = 256 threads, 80 KB of LDS, two workgroups per CU): 1.04 ms per 256^3 against 1.24 ms with
    // 16 x 8 (192 threads; IAMRX_GODUNOV_PTX = 16), 1.14 ms with 30 x 6 -- see the tile shapes of k_god_z below
    const int ptx = (int)tune("GODUNOV_PTX", 14);
    // a domain with walls: the tiles no boundary condition reaches run the plain code (sel = 1), the boundary-condition variant takes the
    // rest (sel = 2: the tiles along the walls, thin first / last z-chunks) -- IAMRX_GODUNOV_SPLIT_BC = 0: one launch of the variant
    const bool split = bcs && tune("GODUNOV_SPLIT_BC", 1) != 0;
#define IAMRX_PZS(TX, TY, B, PPM, SEL) launch_pred_z<TX, TY, B, PPM>(l, vel, force, umac, dP, SEL, &g)
#define IAMRX_PZP(TX, TY, PPM) do { if (!bcs) IAMRX_PZS(TX, TY, false, PPM, 0); else if (!split) IAMRX_PZS(TX, TY, true, PPM, 0); \
                                    else { IAMRX_PZS(TX, TY, false, PPM, 1); IAMRX_PZS(TX, TY, true, PPM, 2); } } while (0)
#define IAMRX_PZ(TX, TY) do { if (ppm) IAMRX_PZP(TX, TY, true); else IAMRX_PZP(TX, TY, false); } while (0)
    if (ptx == 16) IAMRX_PZ(16, 8); else IAMRX_PZ(14, 14);
#undef IAMRX_PZ
#undef IAMRX_PZP
#undef IAMRX_PZS
}


// -------------------------------------------------------------------------------- BDS (ns.advection_scheme = BDS)
// Bell-Dawson-Shubin edge states: the ComputeFluxesOnBoxFromState(..., "BDS") branch of NavierStokesBase::ComputeAofs
// (Source/NavierStokesBase.cpp:4701-4717; AMReX-Hydro's BDS::ComputeEdgeState is not in the reference tree -- restated from the
// published algorithm, Docs/sphinx_documentation/source/TimeStep.rst:92-133, Nonaka et al. SISC 33 (2011); DESIGN.md section 2 for
// the statement of the scheme and tests/test_cpu_bds.py for its known answers).  Three level-wide launches per component: the
// fourth-order nodal interpolant, the limited trilinear slopes (7 per cell, cells grown by one), the edge states of every direction.
namespace {
struct BdsGeom { double h[3]; int dlo[3], dhi[3]; int lo_phys[3], hi_phys[3]; };

__device__ __forceinline__ double bds_eval(double s0, const double* sl, const double* del)
{
    return s0 + del[0] * sl[0] + del[1] * sl[1] + del[2] * sl[2] + del[0] * del[1] * sl[3] + del[0] * del[2] * sl[4] + del[1] * del[2] * sl[5]
              + del[0] * del[1] * del[2] * sl[6];
}
struct BdsCtx {
    FabD s, sl, mac[3], fq;
    int n, conserv, has_f;
    double dt, h[3];
    __device__ __forceinline__ double macv(int d, const int f[3]) const { return mac[d](f[0], f[1], f[2]); }
    __device__ __forceinline__ double dvel(int d, const int q[3]) const
    {
        int p[3] = {q[0], q[1], q[2]};
        p[d] += 1;
        return (macv(d, p) - macv(d, q)) / h[d];
    }
    __device__ __forceinline__ double poly(const int q[3], const double del[3]) const
    {
        double sl7[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) sl7[m] = sl(q[0], q[1], q[2], m);
        return bds_eval(s(q[0], q[1], q[2], n), sl7, del);
    }
};

__device__ double bds_edge(const BdsCtx& c, int D, const int f[3])
{
    const double dt = c.dt, dt2 = dt / 2.0, dt3 = dt / 3.0, dt4 = dt / 4.0;
    const double* h = c.h;
    const double uD = c.macv(D, f);
    const int sgnD = uD > 0.0 ? 1 : -1, offD = uD > 0.0 ? -1 : 0;
    int I[3] = {f[0], f[1], f[2]};
    I[D] += offD;
    double del[3] = {0.0, 0.0, 0.0};
    del[D] = sgnD * 0.5 * h[D] - 0.5 * uD * dt;
    double sedge = c.poly(I, del);
    const int Ta = (D + 1) % 3, Tb = (D + 2) % 3;
    if (c.conserv) sedge *= 1.0 - dt2 * c.dvel(D, I);
    else sedge *= 1.0 + dt2 * (c.dvel(Ta, I) + c.dvel(Tb, I));
    if (c.has_f) sedge += dt2 * c.fq(I[0], I[1], I[2], c.n);
    for (int pass = 0; pass < 2; ++pass) {
        const int T = pass == 0 ? Ta : Tb, O = pass == 0 ? Tb : Ta;
        for (int sideT = 1; sideT >= -1; sideT -= 2) {
            int ft[3] = {I[0], I[1], I[2]};
            if (sideT > 0) ft[T] += 1;
            const double V = c.macv(T, ft);
            const int sgnT = V > 0.0 ? 1 : -1;
            const int offT = sideT > 0 ? (V > 0.0 ? 0 : 1) : (V > 0.0 ? -1 : 0);
            int J[3] = {I[0], I[1], I[2]};
            J[T] += offT;
            int fD[3] = {f[0], f[1], f[2]};
            fD[T] += offT;
            const double uS = c.macv(D, fD);
            const double u = uD * uS > 0.0 ? uS : 0.0;
            double p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0}, p3[3] = {0, 0, 0};
            p1[D] = sgnD * 0.5 * h[D];           p1[T] = sgnT * 0.5 * h[T];
            p2[D] = sgnD * 0.5 * h[D] - uD * dt; p2[T] = sgnT * 0.5 * h[T];
            p3[D] = sgnD * 0.5 * h[D] - u * dt;  p3[T] = sgnT * 0.5 * h[T] - V * dt;
            double d1[3], d2[3], d3[3];
            for (int l = 0; l < 3; ++l) { d1[l] = 0.5 * (p2[l] + p3[l]); d2[l] = 0.5 * (p1[l] + p3[l]); d3[l] = 0.5 * (p1[l] + p2[l]); }
            double gamma = (c.poly(J, d1) + c.poly(J, d2) + c.poly(J, d3)) / 3.0;
            if (c.conserv) gamma *= 1.0 - dt3 * (c.dvel(D, J) + c.dvel(T, J));
            else gamma *= 1.0 + dt3 * c.dvel(O, J);
            for (int sideO = 1; sideO >= -1; sideO -= 2) {
                int fo[3] = {J[0], J[1], J[2]};
                if (sideO > 0) fo[O] += 1;
                const double W = c.macv(O, fo);
                const int sgnO = W > 0.0 ? 1 : -1;
                const int offO = sideO > 0 ? (W > 0.0 ? 0 : 1) : (W > 0.0 ? -1 : 0);
                int K[3] = {J[0], J[1], J[2]};
                K[O] += offO;
                int fD2[3] = {fD[0], fD[1], fD[2]};
                fD2[O] += offO;
                const double uS2 = c.macv(D, fD2);
                const double uu = uD * uS2 > 0.0 ? uS2 : 0.0;
                int ft2[3] = {ft[0], ft[1], ft[2]};
                ft2[O] += offO;
                const double vS2 = c.macv(T, ft2);
                const double vv = V * vS2 > 0.0 ? vS2 : 0.0;
                double q1[3], q2[3], q3[3], q4[3];
                for (int l = 0; l < 3; ++l) { q1[l] = p1[l]; q2[l] = p2[l]; q3[l] = p3[l]; q4[l] = 0.0; }
                q1[O] = q2[O] = q3[O] = sgnO * 0.5 * h[O];
                q4[D] = sgnD * 0.5 * h[D] - uu * dt; q4[T] = sgnT * 0.5 * h[T] - vv * dt; q4[O] = sgnO * 0.5 * h[O] - W * dt;
                double e1[3], e2[3], e3[3], e4[3], e5[3];
                const double a = 0.5, b = 1.0 / 6.0;
                for (int l = 0; l < 3; ++l) {
                    e1[l] = a * q1[l] + b * q2[l] + b * q3[l] + b * q4[l];
                    e2[l] = b * q1[l] + a * q2[l] + b * q3[l] + b * q4[l];
                    e3[l] = b * q1[l] + b * q2[l] + a * q3[l] + b * q4[l];
                    e4[l] = b * q1[l] + b * q2[l] + b * q3[l] + a * q4[l];
                    e5[l] = 0.25 * (q1[l] + q2[l] + q3[l] + q4[l]);
                }
                double gamma2 = -0.8 * c.poly(K, e5) + 0.45 * (c.poly(K, e1) + c.poly(K, e2) + c.poly(K, e3) + c.poly(K, e4));
                if (c.conserv) gamma2 *= 1.0 - dt4 * (c.dvel(D, K) + c.dvel(T, K) + c.dvel(O, K));
                gamma2 *= W;
                gamma -= (double)sideO * dt * gamma2 / (3.0 * h[O]);
            }
            gamma *= V;
            sedge -= (double)sideT * dt * gamma / (2.0 * h[T]);
        }
    }
    return sedge;
}

inline bool bds_is_phys(int b) { return b == bc_foextrap || b == bc_hoextrap || b == bc_ext_dir; }
}  // namespace

// edge[d](.., n) for n < ncomp on the valid faces of every box; S: >= 3 filled ghost cells, umac: >= 1 filled ghost layer
static void bds_edge_states(const Geometry& g, const MultiFab& S, int ncomp, const MultiFab* force, MultiFab* const umac[3], const int* iconserv,
                            double dt, const BCRec* bc, MultiFab* const edge[3])
{
    auto& ctx = Context::get();
    const Layout& l = *S.layout;
    MultiFab sint(S.layout, node_type(), 1, 2), sl(S.layout, cell_type(), 7, 1);
    for (int n = 0; n < ncomp; ++n) {
        BdsGeom G;
        for (int d = 0; d < 3; ++d) {
            G.h[d] = g.dx[d]; G.dlo[d] = g.domain.lo[d]; G.dhi[d] = g.domain.hi[d];
            G.lo_phys[d] = (!g.periodic[d] && bc && bds_is_phys(bc[n].lo[d])) ? 1 : 0;
            G.hi_phys[d] = (!g.periodic[d] && bc && bds_is_phys(bc[n].hi[d])) ? 1 : 0;
        }
        const FabD *st = S.d_tab, *it = sint.d_tab, *lt = sl.d_tab;
        // 1. nodal values: node (i,j,k) = lower corner of cell (i,j,k); tensor product of (-1, 7, 7, -1) / 12 over the 4^3 cells around
        // it; on / beyond a physical boundary: the average of the four ghost cells around the node in the boundary plane
        for_each(l, node_type(), 1, ctx.stream, [=] __device__(int i, int j, int k, int f) {
            const int idx[3] = {i, j, k};
            int bd = -1, bcell = 0;
            for (int d = 0; d < 3 && bd < 0; ++d) {
                if (G.lo_phys[d] && idx[d] <= G.dlo[d]) { bd = d; bcell = G.dlo[d] - 1; }
                else if (G.hi_phys[d] && idx[d] >= G.dhi[d] + 1) { bd = d; bcell = G.dhi[d] + 1; }
            }
            double v = 0.0;
            if (bd >= 0) {
                const int t1 = (bd + 1) % 3, t2 = (bd + 2) % 3;
                for (int b = -1; b <= 0; ++b) for (int a = -1; a <= 0; ++a) {
                    int c[3];
                    c[bd] = bcell; c[t1] = idx[t1] + a; c[t2] = idx[t2] + b;
                    v += 0.25 * st[f](c[0], c[1], c[2], n);
                }
            } else {
                const double w4[4] = {-1.0 / 12.0, 7.0 / 12.0, 7.0 / 12.0, -1.0 / 12.0};
                for (int c = 0; c < 4; ++c) for (int b = 0; b < 4; ++b) for (int a = 0; a < 4; ++a)
                    v += w4[a] * w4[b] * w4[c] * st[f](i - 2 + a, j - 2 + b, k - 2 + c, n);
            }
            it[f](i, j, k) = v;
        });
        // 2. limited slopes on the cells grown by one
        for_each(l, cell_type(), 1, ctx.stream, [=] __device__(int i, int j, int k, int f) {
            const double* h = G.h;
            double sc[8], smin[8], smax[8], sl7[7];
            const double s0 = st[f](i, j, k, n);
            for (int m = 0; m < 8; ++m) sc[m] = it[f](i + (m & 1), j + ((m >> 1) & 1), k + ((m >> 2) & 1));
            auto corners_to_slopes = [&]() {
                sl7[0] = 0.25 * ((sc[1] + sc[3] + sc[5] + sc[7]) - (sc[0] + sc[2] + sc[4] + sc[6])) / h[0];
                sl7[1] = 0.25 * ((sc[2] + sc[3] + sc[6] + sc[7]) - (sc[0] + sc[1] + sc[4] + sc[5])) / h[1];
                sl7[2] = 0.25 * ((sc[4] + sc[5] + sc[6] + sc[7]) - (sc[0] + sc[1] + sc[2] + sc[3])) / h[2];
                sl7[3] = 0.5 * ((sc[0] + sc[3] + sc[4] + sc[7]) - (sc[1] + sc[2] + sc[5] + sc[6])) / (h[0] * h[1]);
                sl7[4] = 0.5 * ((sc[0] + sc[5] + sc[2] + sc[7]) - (sc[1] + sc[4] + sc[3] + sc[6])) / (h[0] * h[2]);
                sl7[5] = 0.5 * ((sc[0] + sc[6] + sc[1] + sc[7]) - (sc[2] + sc[4] + sc[3] + sc[5])) / (h[1] * h[2]);
                sl7[6] = ((sc[7] + sc[1] + sc[2] + sc[4]) - (sc[0] + sc[3] + sc[5] + sc[6])) / (h[0] * h[1] * h[2]);
            };
            corners_to_slopes();
            for (int m = 0; m < 8; ++m) {
                const int mx = m & 1, my = (m >> 1) & 1, mz = (m >> 2) & 1;
                const double del[3] = {(mx ? 0.5 : -0.5) * h[0], (my ? 0.5 : -0.5) * h[1], (mz ? 0.5 : -0.5) * h[2]};
                sc[m] = bds_eval(s0, sl7, del);
                double mn = s0, mxv = s0;
                for (int c = -1; c <= 0; ++c) for (int b = -1; b <= 0; ++b) for (int a = -1; a <= 0; ++a) {
                    const double q = st[f](i + mx + a, j + my + b, k + mz + c, n);
                    mn = fmin(mn, q); mxv = fmax(mxv, q);
                }
                smin[m] = mn; smax[m] = mxv;
                sc[m] = fmax(fmin(sc[m], smax[m]), smin[m]);
            }
            const double eps = 1.0e-10;
            for (int ll = 0; ll < 3; ++ll) {
                double sumloc = 0.0;
                for (int m = 0; m < 8; ++m) sumloc += sc[m];
                sumloc *= 0.125;
                double sumdif = (sumloc - s0) * 8.0;
                const double sgndif = copysign(1.0, sumdif);
                double diff[8];
                int kdp = 0;
                for (int m = 0; m < 8; ++m) { diff[m] = (sc[m] - s0) * sgndif; if (diff[m] > eps) ++kdp; }
                for (int m = 0; m < 8; ++m) {
                    const double div = kdp < 1 ? 1.0 : (double)kdp;
                    double redfac = 0.0;
                    if (diff[m] > eps) { redfac = sumdif * sgndif / div; --kdp; }
                    const double redmax = sgndif > 0.0 ? sc[m] - smin[m] : smax[m] - sc[m];
                    redfac = fmin(redfac, redmax);
                    sumdif -= redfac * sgndif;
                    sc[m] -= redfac * sgndif;
                }
            }
            corners_to_slopes();
            for (int q = 0; q < 7; ++q) lt[f](i, j, k, q) = sl7[q];
        });
        // 3. edge states
        for (int D = 0; D < 3; ++D) {
            const FabD *et = edge[D]->d_tab, *ux = umac[0]->d_tab, *uy = umac[1]->d_tab, *uz = umac[2]->d_tab;
            const FabD* ft = force ? force->d_tab : nullptr;
            const int conserv = iconserv[n] ? 1 : 0, lo_p = G.lo_phys[D], hi_p = G.hi_phys[D];
            for_each(l, face_type(D), 0, ctx.stream, [=] __device__(int i, int j, int k, int f) {
                const int fi[3] = {i, j, k};
                double v;
                if (lo_p && fi[D] == G.dlo[D]) { int c[3] = {i, j, k}; c[D] -= 1; v = st[f](c[0], c[1], c[2], n); }
                else if (hi_p && fi[D] == G.dhi[D] + 1) v = st[f](i, j, k, n);
                else {
                    BdsCtx c;
                    c.s = st[f]; c.sl = lt[f]; c.mac[0] = ux[f]; c.mac[1] = uy[f]; c.mac[2] = uz[f];
                    c.has_f = ft != nullptr; if (ft) c.fq = ft[f]; else c.fq = st[f];
                    c.n = n; c.conserv = conserv; c.dt = dt; c.h[0] = G.h[0]; c.h[1] = G.h[1]; c.h[2] = G.h[2];
                    v = bds_edge(c, D, fi);
                }
                et[f](i, j, k, n) = v;
            });
        }
    }
}

static bool use_tile_kernel()
{
    const int v = (int)tune("GODUNOV_TILE", 0);
    return v != 0;
}

void godunov_compute_aofs(const Geometry& g, MultiFab& aofs, int acomp, const MultiFab& S, int ncomp, const MultiFab* force,
                          const MultiFab* divu, MultiFab* const umac[3], const int* iconserv, double dt, const BCRec* bc,
                          bool is_velocity, bool use_forces_in_trans, MultiFab* const edge_out[3], MultiFab* const flux_out[3], int scheme)
{
    if (S.nlocal() == 0) return;
    IAMRX_ASSERT(S.ngrow >= 3 && ncomp <= GOD_MAXC && S.ncomp >= ncomp);
    IAMRX_ASSERT(umac[0]->ngrow >= 1);
    auto& ctx = Context::get();
    const Layout& l = *S.layout;
    MultiFab e0[3], edge[3], sl[3];
    MultiFab* ed[3];
    const bool ws = use_dir_fused();
    if (scheme == 2) {
        // BDS edge states, then the flux / divergence / convective-term kernel of the Godunov path
        IAMRX_ASSERT(force == nullptr || force->ngrow >= 1);
        MultiFab edge_own[3];
        MultiFab* edp[3];
        for (int d = 0; d < 3; ++d) {
            if (edge_out && edge_out[d]) edp[d] = edge_out[d];
            else { edge_own[d].define(S.layout, face_type(d), ncomp, 0); edp[d] = &edge_own[d]; }
        }
        bds_edge_states(g, S, ncomp, force, umac, iconserv, dt, bc, edp);
        const GodParams* dPb = upload_params(make_params(g, dt, ncomp, bc, iconserv, is_velocity, use_forces_in_trans, force != nullptr, divu != nullptr));
        Tiling tb = level_tiling(l, cell_type(), 0, 4);
        const bool sfb = flux_out && flux_out[0];
        hipLaunchKernelGGL(k_aofs, tb.grid(), Tiling::block(), 0, ctx.stream, tb, l.d_boxes, aofs.d_tab, acomp,
                           edp[0]->d_tab, edp[1]->d_tab, edp[2]->d_tab, umac[0]->d_tab, umac[1]->d_tab, umac[2]->d_tab,
                           sfb ? flux_out[0]->d_tab : nullptr, sfb ? flux_out[1]->d_tab : nullptr, sfb ? flux_out[2]->d_tab : nullptr, dPb);
        return;
    }
    const bool ppm = scheme == 1;
    set_scheme(scheme);
    const bool zk = use_z_kernel();
    for (int d = 0; d < 3 && !use_tile_kernel() && !zk; ++d) {
        e0[d].define(S.layout, face_type(d), ncomp, 1);
        if (ws) sl[d].define(S.layout, cell_type(), ncomp, 1);
        if (edge_out && edge_out[d]) ed[d] = edge_out[d];
        else { edge[d].define(S.layout, face_type(d), ncomp, 0); ed[d] = &edge[d]; }
    }
    const GodParams* dP = upload_params(make_params(g, dt, ncomp, bc, iconserv, is_velocity, use_forces_in_trans, force != nullptr, divu != nullptr));
    if (zk) {
        const int ztx = (int)tune("GODUNOV_ZTX", 14);
        const bool bcs = !(g.periodic[0] && g.periodic[1] && g.periodic[2]);
        // (walls: two launches, see godunov_pred_z)
        const bool split = bcs && tune("GODUNOV_SPLIT_BC", 1) != 0;
#define IAMRX_GZS(TX, TY, W, B, PPM, SEL) launch_god_z<TX, TY, W, B, PPM>(l, aofs, acomp, S, ncomp, force, divu, umac, edge_out, flux_out, dP, SEL, &g)
#define IAMRX_GZP(TX, TY, W, PPM) do { if (!bcs) IAMRX_GZS(TX, TY, W, false, PPM, 0); else if (!split) IAMRX_GZS(TX, TY, W, true, PPM, 0); \
                                       else { IAMRX_GZS(TX, TY, W, false, PPM, 1); IAMRX_GZS(TX, TY, W, true, PPM, 2); } } while (0)
#define IAMRX_GZ(TX, TY, W) do { if (ppm) IAMRX_GZP(TX, TY, W, true); else IAMRX_GZP(TX, TY, W, false); } while (0)
        // Tile shapes, 3 components at 256^3 (periodic, PLM): the 251 VGPRs of the kernel allow 8 wavefronts per CU, so a 192-thread
        // workgroup (16 x 8 cells, grown tile 18 x 10) leaves the CU with 6 wavefronts -- two SIMDs run one -- and 128 useful columns per
        // 192 threads: 2.46 ms.  14 x 14 cells = a grown tile of exactly 16 x 16 = 256 threads: two workgroups = 8 wavefronts per CU, 196
        // useful columns per 256 threads, 53 KB of LDS: 1.84 ms.  30 x 6 (32 x 8 grown): 2.03 ms; 512 threads (30 x 14: 2.25 ms, 14 x 30:
        // 2.33 ms; one workgroup per CU); round 2's 16 x 16 and 32 x 8 cells (384 threads): slower than 16 x 8.
        // IAMRX_GODUNOV_ZTX = 16: the 16 x 8 tiles.
        if (ztx == 16) IAMRX_GZ(16, 8, 2);
        else IAMRX_GZ(14, 14, 2);
#undef IAMRX_GZ
#undef IAMRX_GZP
#undef IAMRX_GZS
        return;
    }
    if (use_tile_kernel()) {
        constexpr int TX = 16, TY = 8, TZ = 4, NTH = 1024;
        const int ntx = (l.max_len[0] + TX - 1) / TX, nty = (l.max_len[1] + TY - 1) / TY, ntz = (l.max_len[2] + TZ - 1) / TZ;
        const bool se = edge_out && edge_out[0], sf = flux_out && flux_out[0];
        dim3 grid((unsigned)(ntx * nty * ntz), (unsigned)l.nlocal());
        hipLaunchKernelGGL((k_godunov_tile<TX, TY, TZ, NTH>), grid, dim3(NTH), 0, ctx.stream, l.d_boxes, S.d_tab, force ? force->d_tab : nullptr,
                           divu ? divu->d_tab : nullptr, umac[0]->d_tab, umac[1]->d_tab, umac[2]->d_tab, aofs.d_tab, acomp,
                           se ? edge_out[0]->d_tab : nullptr, se ? edge_out[1]->d_tab : nullptr, se ? edge_out[2]->d_tab : nullptr,
                           sf ? flux_out[0]->d_tab : nullptr, sf ? flux_out[1]->d_tab : nullptr, sf ? flux_out[2]->d_tab : nullptr,
                           dP, ntx, nty, ntz);
        return;
    }
    launch_trace<false, 0>(l, S, force, *umac[0], e0[0], ws ? &sl[0] : nullptr, dP);
    launch_trace<false, 1>(l, S, force, *umac[1], e0[1], ws ? &sl[1] : nullptr, dP);
    launch_trace<false, 2>(l, S, force, *umac[2], e0[2], ws ? &sl[2] : nullptr, dP);
    launch_final<false, 0>(l, S, force, divu, umac, e0, sl, *ed[0], dP);
    launch_final<false, 1>(l, S, force, divu, umac, e0, sl, *ed[1], dP);
    launch_final<false, 2>(l, S, force, divu, umac, e0, sl, *ed[2], dP);
    Tiling t = level_tiling(l, cell_type(), 0, 4);
    const bool sf = flux_out && flux_out[0];
    hipLaunchKernelGGL(k_aofs, t.grid(), Tiling::block(), 0, ctx.stream, t, l.d_boxes, aofs.d_tab, acomp,
                       ed[0]->d_tab, ed[1]->d_tab, ed[2]->d_tab, umac[0]->d_tab, umac[1]->d_tab, umac[2]->d_tab,
                       sf ? flux_out[0]->d_tab : nullptr, sf ? flux_out[1]->d_tab : nullptr, sf ? flux_out[2]->d_tab : nullptr, dP);
}

}  // namespace iamrx
