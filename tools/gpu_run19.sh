cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in base u2 u7 base; do
  if [ $v = base ]; then unset COSLAM_HIP_LIB; else export COSLAM_HIP_LIB=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip_$v.so; fi
  echo "== $v"; python tools/group_cam.py quick 2>&1 | grep -E "cameras, fused|sampling|total  " | tail -4
done
