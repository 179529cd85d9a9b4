#!/bin/bash
# round 5: k_nodal_gsr under other instruction-scheduling strategies of the AMDGPU back end (compiled on the GPU box, A/B by tools/bench_gsr.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/iamr_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-result -ffp-contract=off"
OBJ="mf.o k_basic.o k_bc.o k_abec.o k_tensor.o k_godunov.o k_nodal.o mlmg.o nodalmg.o macproj.o projection.o diffusion.o navierstokes.o amr.o amrns.o amrregrid.o regrid.o comm.o cabi.o"
cp k_nodal.o /tmp/k_nodal_base.o
run() {
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libiamrx.so $OBJ -ldl && (cd $R && python tools/bench_gsr.py 256 2>&1 | tail -3)
}
echo "== baseline"; run
for f in "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-schedule-relaxed-occupancy=true"; do
    echo "== $f"
    if timeout 600 /opt/rocm/bin/hipcc $BASE $f -c k_nodal.hip -o k_nodal.o 2> /tmp/gsr_flags.err; then run; else tail -2 /tmp/gsr_flags.err; fi
done
cp /tmp/k_nodal_base.o k_nodal.o
