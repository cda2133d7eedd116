cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
for v in "" _u7 _pad _u7pad _wpb4 _u7wpb4 _u2wpb4; do
  echo "=== variant '$v'" | tee -a gpurun_out/r02b/ab.log
  COSLAM_HIP_LIB=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip$v.so timeout 300 python tools/group_cam.py quick >> gpurun_out/r02b/ab.log 2>&1
done
cat gpurun_out/r02b/ab.log
