/* tests/c/mac_shim.c -- a C (not Python) client of libiamrx.so: the call that INTEGRATION.md section 3 puts into
 * MacProj::mlmg_mac_solve (reference Source/MacProj.cpp:1084-1184), driven over caller-owned DEVICE memory through the zero-copy
 * alias (iamrx_mf_alias) exactly as an AMReX GPU build would hand over its FABs.
 *
 * Build (tests/test_gpu_cabi_c.py does this):  gcc mac_shim.c -I../../include -L../../iamr_amd -liamrx -L/opt/rocm/lib -lamdhip64 -lm
 * Exit code 0 = the projected MAC field is discretely divergence free and the caller's buffers were updated in place.
 * Prints:  iters resnorm maxdiv_before maxdiv_after */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "iamrx.h"

/* the three HIP runtime calls a host program needs; declared here so that the test compiles with plain gcc */
extern int hipMalloc(void** p, size_t n);
extern int hipMemcpy(void* dst, const void* src, size_t n, int kind);   /* 1 = H2D, 2 = D2H */
extern int hipFree(void* p);

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "iamrx error: %s (%s:%d)\n", iamrx_last_error(), __FILE__, __LINE__); return 2; } } while (0)

static double* dev_copy(const double* h, size_t n)
{
    void* d = NULL;
    if (hipMalloc(&d, n * sizeof(double)) != 0) return NULL;
    if (hipMemcpy(d, h, n * sizeof(double), 1) != 0) return NULL;
    return (double*)d;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 32;
    const double pi = 3.14159265358979323846, h = 1.0 / n;
    CHECK(iamrx_init(0));
    iamrx_geom g;
    for (int d = 0; d < 3; ++d) { g.dom_lo[d] = 0; g.dom_hi[d] = n - 1; g.prob_lo[d] = 0.0; g.prob_hi[d] = 1.0; g.periodic[d] = 1; }
    const int box[6] = {0, 0, 0, n - 1, n - 1, n - 1}, owner[1] = {0};
    iamrx_layout lay;
    CHECK(iamrx_layout_create(1, box, owner, &lay));
    /* caller-side "FABs": Array4 layout with ghost cells, host images first */
    const int ng = 1;
    size_t nf[3], nc_ = (size_t)(n + 2) * (n + 2) * (n + 2);
    double* hum[3];
    for (int d = 0; d < 3; ++d) {
        const int e[3] = {n + 2 * ng + (d == 0), n + 2 * ng + (d == 1), n + 2 * ng + (d == 2)};
        nf[d] = (size_t)e[0] * e[1] * e[2];
        hum[d] = (double*)calloc(nf[d], sizeof(double));
        for (int k = 0; k < e[2]; ++k) for (int j = 0; j < e[1]; ++j) for (int i = 0; i < e[0]; ++i) {
            /* a periodic field with divergence: u = sin 2pi x, v = cos 2pi y sin 2pi x, w = sin 2pi z (face centres) */
            const double x = (i - ng + (d == 0 ? 0.0 : 0.5)) * h, y = (j - ng + (d == 1 ? 0.0 : 0.5)) * h, z = (k - ng + (d == 2 ? 0.0 : 0.5)) * h;
            const double v = d == 0 ? sin(2 * pi * x) : (d == 1 ? cos(2 * pi * y) * sin(2 * pi * x) : sin(2 * pi * z));
            hum[d][(size_t)i + e[0] * ((size_t)j + (size_t)e[1] * k)] = v;
        }
    }
    double* hrho = (double*)malloc(nc_ * sizeof(double));
    for (size_t q = 0; q < nc_; ++q) hrho[q] = 1.0;
    double* hphi = (double*)calloc(nc_, sizeof(double));
    double *dum[3], *drho = dev_copy(hrho, nc_), *dphi = dev_copy(hphi, nc_);
    iamrx_mf um[3], rho, phi, div;
    for (int d = 0; d < 3; ++d) {
        dum[d] = dev_copy(hum[d], nf[d]);
        const int type[3] = {d == 0, d == 1, d == 2};
        double* ptrs[1] = {dum[d]};
        CHECK(iamrx_mf_alias(lay, type, 1, ng, ptrs, &um[d]));
    }
    const int cell[3] = {0, 0, 0};
    { double* p1[1] = {drho}; CHECK(iamrx_mf_alias(lay, cell, 1, 1, p1, &rho)); }
    { double* p1[1] = {dphi}; CHECK(iamrx_mf_alias(lay, cell, 1, 1, p1, &phi)); }
    CHECK(iamrx_mf_create(lay, cell, 1, 0, &div));
    double d0, d1;
    CHECK(iamrx_mac_divergence(&g, div, um[0], um[1], um[2]));
    CHECK(iamrx_mf_norm0(div, 0, 1, 0, &d0));
    /* the body of MacProj::mlmg_mac_solve: rhs_scale = 2/dt, periodic LinOp BCs, max_order 4 */
    const double dt = 0.01;
    const int lobc[3] = {0, 0, 0}, hibc[3] = {0, 0, 0};
    iamrx_mg_opts o;
    iamrx_mg_default_opts(&o);
    o.maxorder = 4;
    iamrx_mg_stats st;
    CHECK(iamrx_mlmg_mac_solve(&g, um[0], um[1], um[2], rho, 0, NULL, phi, 2.0 / dt, lobc, hibc, 1.e-12, 1.e-16, &o, &st));
    CHECK(iamrx_mac_divergence(&g, div, um[0], um[1], um[2]));
    CHECK(iamrx_mf_norm0(div, 0, 1, 0, &d1));
    CHECK(iamrx_sync());
    /* the caller's own buffer was projected in place: read it back without the library */
    double* back = (double*)malloc(nf[0] * sizeof(double));
    if (hipMemcpy(back, dum[0], nf[0] * sizeof(double), 2) != 0) return 3;
    double changed = 0.0;
    for (size_t q = 0; q < nf[0]; ++q) { const double c = fabs(back[q] - hum[0][q]); if (c > changed) changed = c; }
    printf("%d %.6e %.6e %.6e %.6e\n", st.iters, st.resnorm, d0, d1, changed);
    for (int d = 0; d < 3; ++d) { CHECK(iamrx_mf_destroy(um[d])); hipFree(dum[d]); free(hum[d]); }
    CHECK(iamrx_mf_destroy(rho)); CHECK(iamrx_mf_destroy(phi)); CHECK(iamrx_mf_destroy(div));
    hipFree(drho); hipFree(dphi); free(hrho); free(hphi); free(back);
    CHECK(iamrx_layout_destroy(lay));
    return (st.converged && d1 <= 1.e-9 * d0 && changed > 1.e-3) ? 0 : 1;
}
