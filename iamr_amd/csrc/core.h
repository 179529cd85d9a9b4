// iamr_amd/csrc/core.h -- basic types shared by host drivers and HIP kernels (gfx950 only).
//
// Storage convention = AMReX Array4 as seen at IAMR's seam (e.g. reference
// Source/NavierStokesBase.cpp:4665-4677 `.array(mfi,comp)`): contiguous double,
//   offset(i,j,k,n) = (i-lo0) + n0*((j-lo1) + n1*((k-lo2) + n2*n)),  ghost cells included.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <stdexcept>

namespace iamrx {

struct BoxD {
    int lo[3];
    int hi[3];   // inclusive
    __host__ __device__ int len(int d) const { return hi[d] - lo[d] + 1; }
    __host__ __device__ long npts() const { return (long)len(0) * len(1) * len(2); }
    __host__ __device__ bool ok() const { return hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2]; }
    __host__ __device__ bool contains(int i, int j, int k) const
    {
        return i >= lo[0] && i <= hi[0] && j >= lo[1] && j <= hi[1] && k >= lo[2] && k <= hi[2];
    }
};

// device view of one FAB (all components)
struct FabD {
    double* p;
    int lo[3];
    int n[3];
    long cs;   // component stride = n0*n1*n2
    __host__ __device__ long off(int i, int j, int k) const
    {
        return (long)(i - lo[0]) + (long)n[0] * ((long)(j - lo[1]) + (long)n[1] * (long)(k - lo[2]));
    }
    typedef __attribute__((address_space(1))) double gdouble;
#if defined(__HIP_DEVICE_COMPILE__)
    // device side: the table entry is loaded from memory, so the compiler cannot see that p points to global memory and
    // would emit flat_load/flat_store (which also tie up the LDS counter); say so explicitly -> global_load/global_store
    __device__ gdouble& operator()(int i, int j, int k, int c = 0) const { return ((gdouble*)p)[off(i, j, k) + cs * c]; }
    __device__ gdouble* gp() const { return (gdouble*)p; }      // base pointer in the global address space
#else
    __host__ __device__ double& operator()(int i, int j, int k, int c = 0) const { return p[off(i, j, k) + cs * c]; }
    __host__ __device__ double* gp() const { return p; }
#endif
};

inline BoxD make_box(const int lo[3], const int hi[3])
{
    BoxD b;
    for (int d = 0; d < 3; ++d) { b.lo[d] = lo[d]; b.hi[d] = hi[d]; }
    return b;
}
inline BoxD grow(BoxD b, int ng) { for (int d = 0; d < 3; ++d) { b.lo[d] -= ng; b.hi[d] += ng; } return b; }
inline BoxD grow_dir(BoxD b, int d, int ng) { b.lo[d] -= ng; b.hi[d] += ng; return b; }
inline BoxD convert(BoxD b, const int type[3]) { for (int d = 0; d < 3; ++d) b.hi[d] += type[d]; return b; }
inline BoxD shift(BoxD b, int d, int s) { b.lo[d] += s; b.hi[d] += s; return b; }
inline BoxD intersect(const BoxD& a, const BoxD& b)
{
    BoxD r;
    for (int d = 0; d < 3; ++d) { r.lo[d] = a.lo[d] > b.lo[d] ? a.lo[d] : b.lo[d]; r.hi[d] = a.hi[d] < b.hi[d] ? a.hi[d] : b.hi[d]; }
    return r;
}
inline BoxD coarsen(BoxD b, int r)
{
    auto fl = [r](int i) { return i < 0 ? -((-i + r - 1) / r) : i / r; };
    for (int d = 0; d < 3; ++d) { b.lo[d] = fl(b.lo[d]); b.hi[d] = fl(b.hi[d]); }
    return b;
}
inline BoxD refine(BoxD b, int r)
{
    for (int d = 0; d < 3; ++d) { b.lo[d] *= r; b.hi[d] = (b.hi[d] + 1) * r - 1; }
    return b;
}

// IAMR semantics: errors abort (amrex::Abort).  The library throws; the C-ABI boundary converts
// to a non-zero return code + iamrx_last_error().
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define IAMRX_HIP_CHECK(expr)                                                                              \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess)                                                                              \
            throw ::iamrx::Error(std::string("HIP error ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                                 std::to_string(__LINE__));                                                \
    } while (0)

#define IAMRX_ASSERT(cond)                                                                                  \
    do {                                                                                                    \
        if (!(cond)) throw ::iamrx::Error(std::string("assertion failed: " #cond " at ") + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

// amrex::BCType (mathematical BC per component and face)
enum BCType : int { bc_reflect_odd = -1, bc_int_dir = 0, bc_reflect_even = 1, bc_foextrap = 2, bc_ext_dir = 3, bc_hoextrap = 4 };
// amrex::LinOpBCType
enum LinOpBC : int { lo_periodic = 0, lo_dirichlet = 101, lo_neumann = 102,
                     lo_reflect_odd = 104 /* cell-centred solvers: ghost = -first interior cell (LinOpBCType::reflect_odd, normal velocity at a Symmetry face) */,
                     lo_inflow = 103 /* nodal projection only (LinOpBCType::inflow): Neumann operator, the normal velocity outside the face enters div(u) */ };
// IAMR PhysBCType (reference Source/NS_BC.H)
enum PhysBC : int { phys_interior = 0, phys_inflow = 1, phys_outflow = 2, phys_symmetry = 3, phys_slipwall = 4, phys_noslipwall = 5 };

struct BCRec { int lo[3]; int hi[3]; };

struct Geometry {
    BoxD domain;          // cell-centred index box
    double problo[3], probhi[3], dx[3];
    int periodic[3];
    // nodal solves with Neumann walls: wall nodes carry weight 1/2 in sums and dot products (set by NodalMG)
    int half_lo[3] = {0, 0, 0}, half_hi[3] = {0, 0, 0};
};

}  // namespace iamrx
