#!/bin/bash
# round 5: the Godunov kernels under other instruction-scheduling strategies of the AMDGPU back end (compiled on the GPU box, A/B by tools/bench_godunov.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/iamr_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-result -ffp-contract=fast"
OBJ="mf.o k_basic.o k_bc.o k_abec.o k_tensor.o k_godunov.o k_nodal.o mlmg.o nodalmg.o macproj.o projection.o diffusion.o navierstokes.o amr.o amrns.o amrregrid.o regrid.o comm.o cabi.o"
cp k_godunov.o /tmp/k_godunov_base.o
run() {
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libiamrx.so $OBJ -ldl && (cd $R && python tools/bench_godunov.py 256; python tools/bench_godunov.py 256)
}
echo "== baseline"; run
for f in "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-schedule-relaxed-occupancy=true" "-mllvm -amdgpu-sched-strategy=iterative-minreg"; do
    echo "== $f"
    if timeout 600 /opt/rocm/bin/hipcc $BASE $f -c k_godunov.hip -o k_godunov.o 2> /tmp/god_flags.err; then run; else tail -2 /tmp/god_flags.err; fi
done
cp /tmp/k_godunov_base.o k_godunov.o
