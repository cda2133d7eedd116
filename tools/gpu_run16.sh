cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
for v in base p00 p01 p02 base p00; do
  if [ $v = base ]; then unset COSLAM_HIP_LIB; else export COSLAM_HIP_LIB=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip_$v.so; fi
  python bench.py --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"
  python - <<PY
import json
j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', round(j['value'],1), round(j['ms_per_step'],4), round(j['roofline']['avg_launch_us'],1), j['config']['secondary_cfg2']['camera_frames_per_s'])
PY
done
