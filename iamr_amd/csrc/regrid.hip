// iamr_amd/csrc/regrid.hip -- error estimation and grid generation of one AMR level.
//
//  * derive_mag_vort : NS_derive.cpp:86-264 (dermgvort, non-EB branch): |curl u| from centred differences of the FillPatched
//                      velocity (1 ghost cell)
//  * error_tag       : the AMRErrorTag tests NavierStokes::error_setup builds from amr.refinement_indicators
//                      (Source/NS_error.cpp:10-145): value_greater / value_less / vorticity_greater (threshold x 2^level) /
//                      adjacent_difference_greater, optionally restricted to a RealBox; tags are a cell MultiFab of 0 / 1
//  * cluster_tags    : Berger-Rigoutsos clustering of the tagged cells into boxes (the AmrMesh::MakeNewGrids path of AMReX --
//                      upstream, not in /root/reference; restated: buffer by n_error_buf, coarsen to the blocking factor, recursive
//                      signature cuts at holes / largest inflection until grid_eff is met, refine, chop to max_grid_size).
//                      Host code: the tags of a level are a small integer array.
#include "operators.h"
#include "launch.h"
#include <cstring>
#include <algorithm>
#include <cmath>

namespace iamrx {

void derive_mag_vort(const Geometry& g, MultiFab& out, int ocomp, const MultiFab& vel, int vcomp)
{
    if (out.nlocal() == 0) return;
    IAMRX_ASSERT(vel.ngrow >= 1 && vel.ncomp >= vcomp + 3);
    const FabD *ot = out.d_tab, *vt = vel.d_tab;
    const double idx = 1.0 / g.dx[0], idy = 1.0 / g.dx[1], idz = 1.0 / g.dx[2];
    for_each(*out.layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        const FabD d = vt[f];
        const double vx = 0.5 * (d(i + 1, j, k, vcomp + 1) - d(i - 1, j, k, vcomp + 1)) * idx;
        const double wx = 0.5 * (d(i + 1, j, k, vcomp + 2) - d(i - 1, j, k, vcomp + 2)) * idx;
        const double uy = 0.5 * (d(i, j + 1, k, vcomp) - d(i, j - 1, k, vcomp)) * idy;
        const double wy = 0.5 * (d(i, j + 1, k, vcomp + 2) - d(i, j - 1, k, vcomp + 2)) * idy;
        const double uz = 0.5 * (d(i, j, k + 1, vcomp) - d(i, j, k - 1, vcomp)) * idz;
        const double vz = 0.5 * (d(i, j, k + 1, vcomp + 1) - d(i, j, k - 1, vcomp + 1)) * idz;
        ot[f](i, j, k, ocomp) = sqrt((wy - vz) * (wy - vz) + (uz - wx) * (uz - wx) + (vx - uy) * (vx - uy));
    });
}

// mode: 0 GREATER, 1 LESS, 2 VORT (value * 2^level), 3 GRAD (adjacent difference, needs 1 ghost cell).  Tagged cells are set
// to 1, the others are left alone (several indicators accumulate).  rb_lo / rb_hi: optional RealBox (cell centre inside).
void error_tag(const Geometry& g, MultiFab& tags, const MultiFab& field, int comp, int mode, double value, int level,
               const double* rb_lo, const double* rb_hi)
{
    if (tags.nlocal() == 0) return;
    IAMRX_ASSERT(mode >= 0 && mode <= 3 && (mode != 3 || field.ngrow >= 1));
    const FabD *tt = tags.d_tab, *ft = field.d_tab;
    const double thr = mode == 2 ? value * std::pow(2.0, level) : value;
    const bool has_rb = rb_lo && rb_hi;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (has_rb) for (int d = 0; d < 3; ++d) { lo[d] = rb_lo[d]; hi[d] = rb_hi[d]; }
    const double l0 = lo[0], l1 = lo[1], l2 = lo[2], h0 = hi[0], h1 = hi[1], h2 = hi[2];
    const double p0 = g.problo[0], p1 = g.problo[1], p2 = g.problo[2], dx0 = g.dx[0], dx1 = g.dx[1], dx2 = g.dx[2];
    const int d0 = g.domain.lo[0], d1 = g.domain.lo[1], d2 = g.domain.lo[2];
    for_each(*tags.layout, cell_type(), 0, Context::get().stream, [=] __device__(int i, int j, int k, int f) {
        if (has_rb) {
            const double x = p0 + (i - d0 + 0.5) * dx0, y = p1 + (j - d1 + 0.5) * dx1, z = p2 + (k - d2 + 0.5) * dx2;
            if (x < l0 || x > h0 || y < l1 || y > h1 || z < l2 || z > h2) return;
        }
        const FabD a = ft[f];
        const double v = a(i, j, k, comp);
        bool t;
        if (mode == 0 || mode == 2) t = v >= thr;
        else if (mode == 1) t = v <= thr;
        else {
            double m = fabs(a(i + 1, j, k, comp) - v);
            m = fmax(m, fabs(v - a(i - 1, j, k, comp)));
            m = fmax(m, fabs(a(i, j + 1, k, comp) - v));
            m = fmax(m, fabs(v - a(i, j - 1, k, comp)));
            m = fmax(m, fabs(a(i, j, k + 1, comp) - v));
            m = fmax(m, fabs(v - a(i, j, k - 1, comp)));
            t = m >= thr;
        }
        if (t) tt[f](i, j, k) = 1.0;
    });
}

// ----------------------------------------------------------------------------------------------- clustering (host)
namespace {
struct IBox { int lo[3], hi[3]; long vol() const { return (long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1); } };

struct TagGrid {     // tags on the coarsened (blocking-factor) index space of the domain
    int n[3];
    std::vector<unsigned char> t;
    unsigned char at(int i, int j, int k) const { return t[((size_t)k * n[1] + j) * n[0] + i]; }
};

bool shrink(const TagGrid& T, IBox& b, long& ntag)
{
    int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
    ntag = 0;
    for (int k = b.lo[2]; k <= b.hi[2]; ++k) for (int j = b.lo[1]; j <= b.hi[1]; ++j) for (int i = b.lo[0]; i <= b.hi[0]; ++i)
        if (T.at(i, j, k)) {
            ++ntag;
            const int c[3] = {i, j, k};
            for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], c[d]); hi[d] = std::max(hi[d], c[d]); }
        }
    if (ntag == 0) return false;
    for (int d = 0; d < 3; ++d) { b.lo[d] = lo[d]; b.hi[d] = hi[d]; }
    return true;
}

void br(const TagGrid& T, IBox b, double eff, std::vector<IBox>& out)
{
    long ntag;
    if (!shrink(T, b, ntag)) return;
    if ((double)ntag / (double)b.vol() >= eff) { out.push_back(b); return; }
    // signatures
    std::vector<long> sig[3];
    for (int d = 0; d < 3; ++d) sig[d].assign(b.hi[d] - b.lo[d] + 1, 0);
    for (int k = b.lo[2]; k <= b.hi[2]; ++k) for (int j = b.lo[1]; j <= b.hi[1]; ++j) for (int i = b.lo[0]; i <= b.hi[0]; ++i)
        if (T.at(i, j, k)) { ++sig[0][i - b.lo[0]]; ++sig[1][j - b.lo[1]]; ++sig[2][k - b.lo[2]]; }
    int cut_d = -1, cut = -1;      // cut: the first index of the high part
    // 1) a hole (zero signature) closest to the centre, longest direction first
    {
        double best = 1e300;
        for (int d = 0; d < 3; ++d) {
            const int len = (int)sig[d].size();
            for (int q = 1; q < len - 1; ++q)
                if (sig[d][q] == 0) {
                    const double dist = std::fabs(q - 0.5 * (len - 1)) / len;
                    if (dist < best) { best = dist; cut_d = d; cut = b.lo[d] + q; }
                }
        }
    }
    // 2) the strongest inflection of the signature Laplacian
    if (cut_d < 0) {
        long best = 0;
        for (int d = 0; d < 3; ++d) {
            const int len = (int)sig[d].size();
            if (len < 4) continue;
            std::vector<long> lap(len, 0);
            for (int q = 1; q < len - 1; ++q) lap[q] = sig[d][q + 1] - 2 * sig[d][q] + sig[d][q - 1];
            for (int q = 1; q < len - 2; ++q)
                if ((lap[q] < 0) != (lap[q + 1] < 0) || lap[q] == 0 || lap[q + 1] == 0) {
                    const long mag = std::labs(lap[q + 1] - lap[q]);
                    if (mag > best) { best = mag; cut_d = d; cut = b.lo[d] + q + 1; }
                }
        }
    }
    // 3) bisect the longest direction
    if (cut_d < 0) {
        int dl = 0;
        for (int d = 1; d < 3; ++d) if (sig[d].size() > sig[dl].size()) dl = d;
        if (sig[dl].size() < 2) { out.push_back(b); return; }
        cut_d = dl; cut = b.lo[dl] + (int)sig[dl].size() / 2;
    }
    IBox a = b, c = b;
    a.hi[cut_d] = cut - 1; c.lo[cut_d] = cut;
    br(T, a, eff, out);
    br(T, c, eff, out);
}
}  // namespace

// tags_host: domain-sized 0/1 array (x fastest); returns boxes of the SAME index space as the tags (the caller refines them by the
// refinement ratio), each aligned to blocking_factor, at most max_grid_size long, disjoint, covering every tagged cell grown by n_error_buf
std::vector<BoxD> cluster_tags(const unsigned char* tags_host, const BoxD& domain, int blocking_factor, int max_grid_size, double grid_eff,
                               int n_error_buf, const OutflowTags* oft, const unsigned char* allowed)
{
    const int bf = std::max(1, blocking_factor);
    int n[3], nc[3];
    for (int d = 0; d < 3; ++d) {
        n[d] = domain.len(d);
        IAMRX_ASSERT(n[d] % bf == 0 && max_grid_size % bf == 0);
        nc[d] = n[d] / bf;
    }
    TagGrid T;
    for (int d = 0; d < 3; ++d) T.n[d] = nc[d];
    T.t.assign((size_t)nc[0] * nc[1] * nc[2], 0);
    // A tagged cell, buffered by n_error_buf cells (clipped at the domain: periodic wrap of the buffer is left to the proper-nesting step of a
    // multi-level driver) and coarsened by the blocking factor, marks the blocks [(c - buf) / bf, (c + buf) / bf] of every direction: the
    // block ranges of k and j once per row, of i per tagged cell (round 6: the (2 buf + 1)^3 cells of every tagged cell, three divisions
    // each, were 0.2 s per regrid of a 512^3 index space with 10^7 tags)
    std::vector<int> blo[3], bhi[3];
    for (int d = 0; d < 3; ++d) {
        blo[d].resize(n[d]); bhi[d].resize(n[d]);
        for (int c = 0; c < n[d]; ++c) { blo[d][c] = std::max(0, c - n_error_buf) / bf; bhi[d][c] = std::min(n[d] - 1, c + n_error_buf) / bf; }
    }
    for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) {
        const unsigned char* row = tags_host + ((size_t)k * n[1] + j) * n[0];
        for (int i = 0; i < n[0]; ++i) {
            // (most of a level's index space carries no tag: eight cells per test)
            if ((i & 7) == 0 && i + 8 <= n[0]) { unsigned long long w; std::memcpy(&w, row + i, 8); if (w == 0) { i += 7; continue; } }
            if (!row[i]) continue;
            for (int kb = blo[2][k]; kb <= bhi[2][k]; ++kb)
                for (int jb = blo[1][j]; jb <= bhi[1][j]; ++jb) {
                    unsigned char* tr = &T.t[((size_t)kb * nc[1] + jb) * nc[0]];
                    for (int ib = blo[0][i]; ib <= bhi[0][i]; ++ib) tr[ib] = 1;
                }
        }
    }
    // NavierStokesBase::manual_tags_placement (NavierStokesBase.cpp:2112-2215), on the tags coarsened by the blocking factor
    if (oft) for (int f = 0; f < oft->nface; ++f) {
        const int D = oft->dir[f], hi = oft->side[f];
        IBox ob;
        for (int d = 0; d < 3; ++d) { ob.lo[d] = 0; ob.hi[d] = nc[d] - 1; }
        if (oft->mode == 1) {                           // do_refine_outflow: the layer next to the face, all of it if any of it is tagged
            ob.lo[D] = ob.hi[D] = hi ? nc[D] - 1 : 0;
            bool has = false;
            for (int k = ob.lo[2]; k <= ob.hi[2] && !has; ++k) for (int j = ob.lo[1]; j <= ob.hi[1] && !has; ++j) for (int i = ob.lo[0]; i <= ob.hi[0]; ++i)
                if (T.t[((size_t)k * nc[1] + j) * nc[0] + i]) { has = true; break; }
            if (!has) continue;
            for (int k = ob.lo[2]; k <= ob.hi[2]; ++k) for (int j = ob.lo[1]; j <= ob.hi[1]; ++j) for (int i = ob.lo[0]; i <= ob.hi[0]; ++i) T.t[((size_t)k * nc[1] + j) * nc[0] + i] = 1;
        } else if (oft->mode == 2 && oft->ncoarse > 0) {   // do_derefine_outflow: N_coarse_cells layers next to the face are cleared
            if (hi) ob.lo[D] = std::max(0, nc[D] - oft->ncoarse); else ob.hi[D] = std::min(nc[D] - 1, oft->ncoarse - 1);
            for (int k = ob.lo[2]; k <= ob.hi[2]; ++k) for (int j = ob.lo[1]; j <= ob.hi[1]; ++j) for (int i = ob.lo[0]; i <= ob.hi[0]; ++i) T.t[((size_t)k * nc[1] + j) * nc[0] + i] = 0;
        }
    }
    // proper nesting of a regrid that starts above level 0 (Amr::regrid(lbase > 0): the tags outside the proper nesting domain are removed
    // and the clusters are intersected with it): `allowed` marks the cells of this index space the new level may cover; a block of the
    // blocking-factor lattice counts only if all of its cells are allowed
    std::vector<unsigned char> okb;
    if (allowed) {
        okb.assign(T.t.size(), 1);
        for (int k = 0; k < n[2]; ++k) for (int j = 0; j < n[1]; ++j) {
            const unsigned char* row = allowed + ((size_t)k * n[1] + j) * n[0];
            unsigned char* ob = &okb[((size_t)(k / bf) * nc[1] + j / bf) * nc[0]];
            for (int i = 0; i < n[0]; ++i) {
                // (eight cells per test where all of them are allowed; blocks that are already out need no second look)
                if ((i & 7) == 0 && i + 8 <= n[0]) { unsigned long long w; std::memcpy(&w, row + i, 8); if (w == 0x0101010101010101ull) { i += 7; continue; } }
                if (!row[i]) ob[i / bf] = 0;
            }
        }
        for (size_t q = 0; q < T.t.size(); ++q) if (!okb[q]) T.t[q] = 0;
    }
    std::vector<IBox> cl;
    IBox all;
    for (int d = 0; d < 3; ++d) { all.lo[d] = 0; all.hi[d] = nc[d] - 1; }
    br(T, all, grid_eff, cl);
    if (allowed) {                       // a cluster that reaches over blocks that are not allowed: its allowed blocks, row by row
        std::vector<IBox> cl2;
        for (const IBox& c : cl) {
            bool whole = true;
            for (int k = c.lo[2]; k <= c.hi[2] && whole; ++k) for (int j = c.lo[1]; j <= c.hi[1] && whole; ++j) for (int i = c.lo[0]; i <= c.hi[0]; ++i)
                if (!okb[((size_t)k * nc[1] + j) * nc[0] + i]) { whole = false; break; }
            if (whole) { cl2.push_back(c); continue; }
            for (int k = c.lo[2]; k <= c.hi[2]; ++k) for (int j = c.lo[1]; j <= c.hi[1]; ++j) {
                int i = c.lo[0];
                while (i <= c.hi[0]) {
                    if (!okb[((size_t)k * nc[1] + j) * nc[0] + i]) { ++i; continue; }
                    int e = i;
                    while (e + 1 <= c.hi[0] && okb[((size_t)k * nc[1] + j) * nc[0] + e + 1]) ++e;
                    IBox r; r.lo[0] = i; r.hi[0] = e; r.lo[1] = r.hi[1] = j; r.lo[2] = r.hi[2] = k;
                    cl2.push_back(r);
                    i = e + 1;
                }
            }
        }
        cl.swap(cl2);
    }
    // refine to the tag index space and chop to max_grid_size
    std::vector<BoxD> out;
    const int mg = max_grid_size;
    for (const IBox& c : cl) {
        int lo[3], hi[3];
        for (int d = 0; d < 3; ++d) { lo[d] = domain.lo[d] + c.lo[d] * bf; hi[d] = domain.lo[d] + (c.hi[d] + 1) * bf - 1; }
        for (int k0 = lo[2]; k0 <= hi[2]; k0 += mg) for (int j0 = lo[1]; j0 <= hi[1]; j0 += mg) for (int i0 = lo[0]; i0 <= hi[0]; i0 += mg) {
            BoxD b;
            b.lo[0] = i0; b.lo[1] = j0; b.lo[2] = k0;
            b.hi[0] = std::min(i0 + mg - 1, hi[0]); b.hi[1] = std::min(j0 + mg - 1, hi[1]); b.hi[2] = std::min(k0 + mg - 1, hi[2]);
            out.push_back(b);
        }
    }
    return out;
}

}  // namespace iamrx
