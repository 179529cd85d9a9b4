"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the CPU oracle.
TaylorGreen 16^3 (3-D, prob.c = 1): init projections + pressure iterations + 1 full NavierStokes::advance."""
import ctypes as C
import numpy as np


def run():
    import orc
    from iamr_amd import lib
    from iamr_amd import ns as N
    orc.build()
    L = orc.lib()
    lib.init(0)
    n = (16, 16, 16)
    visc = 1e-2
    # oracle
    g = orc.geom(n)
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    p.cfl = 0.5
    p.visc_coef = visc
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    L.orc_ns_init_taylorgreen(s, C.c_double(1.0), C.c_double(1.0), C.c_double(1.0), C.c_double(1.0), C.c_double(1.0))
    L.orc_ns_post_init(s, C.c_double(-1.0))
    L.orc_ns_step(s)
    S_o = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    L.orc_ns_destroy(s)
    # HIP path through the C-ABI
    gd = lib.Geom.make(n)
    ns = N.NavierStokes(gd, lib.Layout.single(n), N.ns_params(cfl=0.5, visc_coef=visc))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    ns.step()
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    err = np.abs(S - S_o).max()
    assert err <= 1e-8, f"smoke: HIP path deviates from the oracle by {err}"
    print(f"smoke ok: max |HIP - oracle| = {err:.3e} after init + 1 step on {n}")


if __name__ == "__main__":
    run()
