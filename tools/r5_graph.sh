cd $GRAFT_REPO_ROOT
for g in 0 1 0 1; do echo "MG_GRAPH=$g"; IAMRX_MG_GRAPH=$g python tools/run_steps.py 2>&1 | grep "ms/step"; done
echo "--- 128"
for g in 0 1; do echo "MG_GRAPH=$g"; IAMRX_N=128 IAMRX_MG_GRAPH=$g python tools/run_steps.py 2>&1 | grep "ms/step"; done
echo "--- ldc"
for g in 0 1; do echo "MG_GRAPH=$g"; IAMRX_MG_GRAPH=$g python tools/run_ldc.py 2>&1 | tail -1; done
