// iamr_amd/csrc/k_bc.hip -- physical boundary fill of cell-centred ghost cells outside the domain.
//
// Role: the BCRec-driven part of FillPatch (amrex FilccCell semantics: int_dir / ext_dir / foextrap / hoextrap /
// reflect_even / reflect_odd) plus IAMR's ext-Dirichlet functors with constant boundary values
// (reference Source/NS_bcfill.H:17-95, BC tables Source/NS_BC.H:7-55, values Source/NavierStokes.cpp:72-83,125-168).
// Directions are applied one after the other (x, then y, then z) over the WHOLE grown extent of the other
// directions, so edge/corner ghosts get the composition of the two/three one-dimensional rules.
#include "kernels.h"
#include "launch.h"
#include <vector>
#include <cstring>

namespace iamrx {

struct PhysBcDesc { int fab; BoxD region; int side; };
struct PhysBcParams {
    int dir, dlo, dhi;
    int ncomp, scomp;
    int bclo[8], bchi[8];
    double edlo[8], edhi[8];
};

__global__ void __launch_bounds__(256) k_physbc(const PhysBcDesc* __restrict__ descs, const FabD* __restrict__ tab, PhysBcParams P)
{
    const PhysBcDesc bd = descs[blockIdx.y];
    const FabD a = tab[bd.fab];
    const int nx = bd.region.len(0), ny = bd.region.len(1);
    const long npts = bd.region.npts();
    const int d = P.dir;
    const int fhi_d = a.lo[d] + a.n[d] - 1;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        int idx[3];
        idx[0] = bd.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        idx[1] = bd.region.lo[1] + (int)(r % ny);
        idx[2] = bd.region.lo[2] + (int)(r / ny);
        const int c = idx[d];
        for (int n = 0; n < P.ncomp; ++n) {
            const int bc = bd.side == 0 ? P.bclo[n] : P.bchi[n];
            int s1[3] = {idx[0], idx[1], idx[2]}, s2[3] = {idx[0], idx[1], idx[2]}, s3[3] = {idx[0], idx[1], idx[2]};
            double v;
            if (bd.side == 0) {
                if (bc == bc_foextrap) { s1[d] = P.dlo; v = a(s1[0], s1[1], s1[2], P.scomp + n); }
                else if (bc == bc_hoextrap) {
                    s1[d] = P.dlo; s2[d] = P.dlo + 1; s3[d] = P.dlo + 2;
                    if (c < P.dlo - 1) v = a(s1[0], s1[1], s1[2], P.scomp + n);
                    else if (P.dlo + 2 <= (fhi_d < P.dhi ? fhi_d : P.dhi))
                        v = 0.125 * (15. * a(s1[0], s1[1], s1[2], P.scomp + n) - 10. * a(s2[0], s2[1], s2[2], P.scomp + n) + 3. * a(s3[0], s3[1], s3[2], P.scomp + n));
                    else v = 0.5 * (3. * a(s1[0], s1[1], s1[2], P.scomp + n) - a(s2[0], s2[1], s2[2], P.scomp + n));
                }
                else if (bc == bc_reflect_even) { s1[d] = 2 * P.dlo - c - 1; v = a(s1[0], s1[1], s1[2], P.scomp + n); }
                else if (bc == bc_reflect_odd) { s1[d] = 2 * P.dlo - c - 1; v = -a(s1[0], s1[1], s1[2], P.scomp + n); }
                else if (bc == bc_ext_dir) v = P.edlo[n];
                else continue;
            } else {
                const int flo_d = a.lo[d];
                if (bc == bc_foextrap) { s1[d] = P.dhi; v = a(s1[0], s1[1], s1[2], P.scomp + n); }
                else if (bc == bc_hoextrap) {
                    s1[d] = P.dhi; s2[d] = P.dhi - 1; s3[d] = P.dhi - 2;
                    if (c > P.dhi + 1) v = a(s1[0], s1[1], s1[2], P.scomp + n);
                    else if (P.dhi - 2 >= (flo_d > P.dlo ? flo_d : P.dlo))
                        v = 0.125 * (15. * a(s1[0], s1[1], s1[2], P.scomp + n) - 10. * a(s2[0], s2[1], s2[2], P.scomp + n) + 3. * a(s3[0], s3[1], s3[2], P.scomp + n));
                    else v = 0.5 * (3. * a(s1[0], s1[1], s1[2], P.scomp + n) - a(s2[0], s2[1], s2[2], P.scomp + n));
                }
                else if (bc == bc_reflect_even) { s1[d] = 2 * P.dhi - c + 1; v = a(s1[0], s1[1], s1[2], P.scomp + n); }
                else if (bc == bc_reflect_odd) { s1[d] = 2 * P.dhi - c + 1; v = -a(s1[0], s1[1], s1[2], P.scomp + n); }
                else if (bc == bc_ext_dir) v = P.edhi[n];
                else continue;
            }
            a(idx[0], idx[1], idx[2], P.scomp + n) = v;
        }
    }
}

// bc[n], extdir_lo/hi[n*3+d] for the ncomp components starting at scomp (ncomp <= 8)
void fill_physbc_cc(const Geometry& g, MultiFab& mf, int scomp, int ncomp, const BCRec* bc, const double* extdir_lo, const double* extdir_hi)
{
    if (mf.nlocal() == 0) return;      // ngrow may be 0: patch arrays (coarse-fine interpolation) extend beyond the domain as such
    IAMRX_ASSERT(ncomp <= 8 && mf.type.cell());
    auto& ctx = Context::get();
    for (int d = 0; d < 3; ++d) {
        if (g.periodic[d]) continue;
        static auto& cache = make_desc_cache<PhysBcDesc>();
        int nd; long maxpts;
        const PhysBcDesc* dd = cached_descs(cache, {(long)mf.layout->id, mf.ngrow, d, g.domain.lo[d], g.domain.hi[d], 0, 0, 0, 0, 0},
            [&](std::vector<PhysBcDesc>& descs, long& mp) {
                for (int li = 0; li < mf.nlocal(); ++li) {
                    const BoxD fb = mf.fabbox(li);
                    for (int side = 0; side < 2; ++side) {
                        BoxD r = fb;
                        if (side == 0) { if (fb.lo[d] >= g.domain.lo[d]) continue; r.hi[d] = g.domain.lo[d] - 1; }
                        else { if (fb.hi[d] <= g.domain.hi[d]) continue; r.lo[d] = g.domain.hi[d] + 1; }
                        descs.push_back({li, r, side});
                        mp = std::max(mp, r.npts());
                    }
                }
            }, nd, maxpts);
        if (nd == 0) continue;
        PhysBcParams P;
        P.dir = d; P.dlo = g.domain.lo[d]; P.dhi = g.domain.hi[d]; P.ncomp = ncomp; P.scomp = scomp;
        for (int n = 0; n < 8; ++n) {
            P.bclo[n] = n < ncomp ? bc[n].lo[d] : 0; P.bchi[n] = n < ncomp ? bc[n].hi[d] : 0;
            P.edlo[n] = (n < ncomp && extdir_lo) ? extdir_lo[n * 3 + d] : 0.0;
            P.edhi[n] = (n < ncomp && extdir_hi) ? extdir_hi[n * 3 + d] : 0.0;
        }
        long nb = (maxpts + 255) / 256; if (nb > 128) nb = 128; if (nb < 1) nb = 1;
        hipLaunchKernelGGL(k_physbc, dim3((unsigned)nb, (unsigned)nd), dim3(256), 0, ctx.stream, dd, mf.d_tab, P);
    }
}

// ------------------------------------------------------------------------------------------------------
// Nodal data at Neumann walls: even reflection of the ghost nodes about the wall node, x(w - m) = x(w + m)
// (mlndlap_applybc).  Directions one after the other over the full grown extent of the others.
struct ReflDesc { int fab; BoxD region; };
struct ReflWalls { int lo[3], hi[3]; int has_lo[3], has_hi[3]; };   // wall node index per direction and side (has_*: Neumann wall there)

// One launch for all directions: a ghost node outside one or more walls takes the value of the node obtained by reflecting
// EVERY out-of-wall coordinate (the same result as reflecting direction by direction over the full extent of the others, since a
// reflected coordinate always lands inside the walls).  Slabs of different directions overlap along edges: they write the same value.
__global__ void __launch_bounds__(256) k_nodal_reflect(const ReflDesc* __restrict__ descs, const FabD* __restrict__ tab, ReflWalls W, int ncomp)
{
    const ReflDesc bd = descs[blockIdx.y];
    const FabD a = tab[bd.fab];
    const int nx = bd.region.len(0), ny = bd.region.len(1);
    const long npts = bd.region.npts();
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += (long)gridDim.x * 256) {
        int idx[3];
        idx[0] = bd.region.lo[0] + (int)(q % nx);
        const long r = q / nx;
        idx[1] = bd.region.lo[1] + (int)(r % ny);
        idx[2] = bd.region.lo[2] + (int)(r / ny);
        int s[3] = {idx[0], idx[1], idx[2]};
        for (int d = 0; d < 3; ++d) {
            if (W.has_lo[d] && idx[d] < W.lo[d]) s[d] = 2 * W.lo[d] - idx[d];
            else if (W.has_hi[d] && idx[d] > W.hi[d]) s[d] = 2 * W.hi[d] - idx[d];
        }
        for (int n = 0; n < ncomp; ++n) a(idx[0], idx[1], idx[2], n) = a(s[0], s[1], s[2], n);
    }
}

void nodal_reflect_bc(const Geometry& g, MultiFab& mf, const DomainBC& bc, hipStream_t on)
{
    if (mf.nlocal() == 0 || mf.ngrow == 0) return;
    auto& ctx = Context::get();
    ReflWalls W;
    long code = 0;
    for (int d = 0; d < 3; ++d) {
        W.lo[d] = g.domain.lo[d]; W.hi[d] = g.domain.hi[d] + 1;      // wall node indices
        W.has_lo[d] = (!g.periodic[d] && bc.lo[d] == lo_neumann) ? 1 : 0;
        W.has_hi[d] = (!g.periodic[d] && bc.hi[d] == lo_neumann) ? 1 : 0;
        code = code * 4 + W.has_lo[d] * 2 + W.has_hi[d];
    }
    if (code == 0) return;
    static auto& cache = make_desc_cache<ReflDesc>();
    int nd; long maxpts;
    const ReflDesc* dd = cached_descs(cache, {(long)mf.layout->id, mf.ngrow, code, g.domain.lo[0], g.domain.lo[1], g.domain.lo[2], g.domain.hi[0],
                                              g.domain.hi[1], g.domain.hi[2], mf.type.t[0] + 2 * mf.type.t[1] + 4 * mf.type.t[2]},
        [&](std::vector<ReflDesc>& descs, long& mp) {
            for (int li = 0; li < mf.nlocal(); ++li) {
                const BoxD fb = mf.fabbox(li);
                for (int d = 0; d < 3; ++d) {
                    if (W.has_lo[d] && fb.lo[d] < W.lo[d]) { BoxD r = fb; r.hi[d] = W.lo[d] - 1; descs.push_back({li, r}); mp = std::max(mp, r.npts()); }
                    if (W.has_hi[d] && fb.hi[d] > W.hi[d]) { BoxD r = fb; r.lo[d] = W.hi[d] + 1; descs.push_back({li, r}); mp = std::max(mp, r.npts()); }
                }
            }
        }, nd, maxpts);
    if (nd == 0) return;
    long nb = (maxpts + 255) / 256; if (nb > 128) nb = 128; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_nodal_reflect, dim3((unsigned)nb, (unsigned)nd), dim3(256), 0, on ? on : ctx.stream, dd, mf.d_tab, W, mf.ncomp);
}

// cell-centred mirror across every non-periodic wall (mlndlap_fillbc_cc for sigma): reflect_even on all faces
void cc_mirror_bc(const Geometry& g, MultiFab& mf)
{
    BCRec bc[8];
    for (int n = 0; n < 8; ++n) for (int d = 0; d < 3; ++d) bc[n].lo[d] = bc[n].hi[d] = bc_reflect_even;
    fill_physbc_cc(g, mf, 0, mf.ncomp, bc, nullptr, nullptr);
}

}  // namespace iamrx
