cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02q; mkdir -p $O
run() { # name lib streamprio
  if [ "$2" = base ]; then unset COSLAM_HIP_LIB; else export COSLAM_HIP_LIB=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip_$2.so; fi
  COSLAM_BA_STREAM_PRIO=$3 python bench.py --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err; echo "$1 rc=$?"
  python - <<PY
import json
j=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
print('$1', round(j['value'],1), round(j['ms_per_step'],4), round(j['roofline']['avg_launch_us'],1))
PY
}
run base base 0
run sp base 1
run bp3 bp3 0
run bp3sp bp3 1
run bp2 bp2 0
run base2 base 0
