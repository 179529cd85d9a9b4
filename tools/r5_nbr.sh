export NAMELEN=70 NTOP=26
echo "=== 8x128 kept"; IAMRX_COALESCE=0 IAMRX_MAXGRID=128 bash tools/profile_step.sh
echo "=== 8 x 256x128x64 kept"; IAMRX_COALESCE=0 IAMRX_MAXGRID=256,128,64 bash tools/profile_step.sh
echo "=== 8 x 256^3 kept (256x512x1024)"; IAMRX_COALESCE=0 IAMRX_N=256,512,1024 IAMRX_MAXGRID=256 bash tools/profile_step.sh
