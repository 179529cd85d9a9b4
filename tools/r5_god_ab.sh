cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python tools/bench_godunov.py; done
cd iamr_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-result -ffp-contract=fast -DIAMRX_GOD_ROW16=0 -c k_godunov.hip -o k_godunov.o && make 2>&1 | tail -1
cd ../..
echo "--- ROW16=0"
for i in 1 2 3; do python tools/bench_godunov.py; done
