"""where the host synchronisations of a time step come from: runs 2 TaylorGreen steps (n^3, default 64) with IAMRX_SYNC_TRACE, resolves the
caller addresses with addr2line and prints the call sites by count per step (scratch tool)"""
import sys, os, subprocess, collections, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from iamr_amd import lib, ns as N
    lib.init(0)
    n = int(sys.argv[2])
    g = lib.Geom.make((n,) * 3); lay = lib.Layout.single((n,) * 3)
    if os.environ.get("TRACE_AMR"):
        # the bench's 2-level workload (n^3 base + n^3 refined box): the events of two coarse steps
        from iamr_amd.amr import Amr
        lays = [lay, lib.Layout([((n // 2,) * 3, (n // 2 + n - 1,) * 3)])]
        amr = Amr(g, lays, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
        for l in range(2): amr.levels[l].init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
        amr.post_init(); amr.coarse_step(); lib.sync()
        class S:
            def step(self): amr.coarse_step()
        s = S()
    else:
        s = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
        s.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0); s.post_init(-1.0); s.step(); lib.sync()
    key = os.environ.get("TRACE_KEY", "SYNC_TRACE")
    lib.tuning_set(key, 1)
    for _ in range(2): s.step()
    lib.tuning_set(key, 0)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "64"
r = subprocess.run([sys.executable, __file__, "child", n], capture_output=True, text=True)
so = os.path.join(ROOT, "iamr_amd", "libiamrx.so")
cnt = collections.Counter()
pts = collections.Counter()      # blas sites: points touched (array points x components), summed
big = int(n) ** 3 // 2
for line in r.stderr.splitlines():
    if line.startswith("iamrx sync:"):
        cnt[tuple(line.split()[2:5])] += 1
    elif line.startswith("iamrx blas:"):
        w = line.split()
        if int(w[3]) >= big:                      # finest-level arrays only
            cnt[(w[2],) + tuple(w[4:7])] += 1
            pts[(w[2],) + tuple(w[4:7])] += int(w[3])
addrs = sorted({a for k in cnt for a in k if re.fullmatch(r"[0-9a-f]+", a) and len(a) > 3})
res = subprocess.run(["addr2line", "-f", "-C", "-e", so] + ["0x" + a for a in addrs], capture_output=True, text=True).stdout.splitlines()
name = {a: re.sub(r"\(.*", "", res[2 * i]).replace("iamrx::", "") + ":" + res[2 * i + 1].split(":")[-1].split()[0] for i, a in enumerate(addrs)}
name.update({k: k for k in ("setVal", "Copy", "saxpy", "lincomb", "mult")})
print("events per step:", sum(cnt.values()) / 2)
if os.environ.get("TRACE_KEY") == "BLAS_TRACE":       # by points touched (in units of one finest-level array), not by count
    unit = 2.0 * int(n) ** 3
    print("array passes per step (points / n^3):", sum(pts.values()) / unit)
    for k, v in pts.most_common():
        print("%6.2f  (%4.1f x)  %s" % (v / unit, cnt[k] / 2, " <- ".join(name[a] for a in k)))
else:
    for k, v in cnt.most_common():
        print("%5.1f  %s" % (v / 2, " <- ".join(name[a] for a in k)))
