export NAMELEN=70 NTOP=45
echo "=== 8x128 kept"; IAMRX_COALESCE=0 IAMRX_MAXGRID=128 bash tools/profile_step.sh
echo "=== 8x256 kept (512^3)"; IAMRX_COALESCE=0 IAMRX_N=512 IAMRX_MAXGRID=256 bash tools/profile_step.sh
echo "=== 512 merged"; IAMRX_N=512 bash tools/profile_step.sh
