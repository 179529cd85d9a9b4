"""The bench's 2-level AMR workload (128^3 base + one 128^3 refined box, nu = 1e-4) between two marker launches: post_init + 1 warm-up
coarse step, then 4 coarse steps -- the workload of `DBG=run_amr_steps.py bash tools/profile_step.sh`."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as N
from iamr_amd.amr import Amr
lib.init(0)
n0 = int(os.environ.get("AMR_N0", "128"))
g0 = lib.Geom.make((n0,) * 3)
lo, hi = n0 // 2, n0 // 2 + n0 - 1
lays = [lib.Layout.single((n0,) * 3), lib.Layout([((lo,) * 3, (hi,) * 3)])]
amr = Amr(g0, lays, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
for l in range(2):
    amr.levels[l].init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
amr.post_init()
amr.coarse_step()
lib.sync()
print("MARK_BEGIN", flush=True)
m = lib.MultiFab(lib.Layout.single((7, 7, 7)), lib.CELL, 1, 0)
m.setval(1.0); m.setval(1.0); m.setval(1.0); lib.sync()   # marker: three consecutive tiny fills
import ctypes as C
if os.environ.get("AMR_SCOPES"):      # scope profile (ProfScope: the stream is drained at both ends of every scope) of the four steps
    lib.check(lib.lib().iamrx_scope_profile(1, 1, None, C.c_size_t(0)))
t0 = time.perf_counter()
for _ in range(4): amr.coarse_step()
lib.sync()
print("ms/step", (time.perf_counter() - t0) * 250)
st, stm = amr.sync_stats()
print("sync project iterations of the last coarse step", st.iters, "residual", st.resnorm, "of", st.resnorm0, "; mac sync iterations", stm.iters)
if os.environ.get("AMR_SCOPES"):
    buf = C.create_string_buffer(1 << 16)
    lib.check(lib.lib().iamrx_scope_profile(0, 0, buf, C.c_size_t(1 << 16)))
    print("scope profile (four coarse steps):")
    print(buf.value.decode())
m.setval(2.0); m.setval(2.0); m.setval(2.0); lib.sync()
