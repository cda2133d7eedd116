# round 3, sweep 2: the changed tests + CU-range partitions for the key-frame solves
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s2; mkdir -p $O
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_comm_gpu.py tests/test_klt_gpu.py tests/test_configs_gpu.py tests/test_sliced_ba_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
run() { # name, args...
  n=$1; shift
  python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 "$@" > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j['roofline']; c=j['config']
    print(f"{sys.argv[2]:28s} {j['value']:8.1f} frames/s  tracker {r['avg_launch_us']:.1f} us x {r['launches_per_frame']} launches  joint steps {c['joint_ba_last']['lm_steps']} ic {c['intercam_last']['lm_steps']}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run base
run k192_ba64 --klt-cus 192 --ba-cus 192:64 --ic-cus 192:64
run k192_ba48_ic16 --klt-cus 192 --ba-cus 192:48 --ic-cus 240:16
run k192_ba64_pose192 --klt-cus 192 --ba-cus 192:64 --ic-cus 192:64 --pose-cus 0:192
run k224_ba32 --klt-cus 224 --ba-cus 224:32 --ic-cus 224:32
run k160_ba96 --klt-cus 160 --ba-cus 160:96 --ic-cus 160:96
run cams4_ba64 --klt-cams-per-launch 4 --ba-cus 192:64 --ic-cus 192:64
cd /tmp
v=k192_ba64
rm -rf /tmp/kt_$v && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --klt-cus 192 --ba-cus 192:64 --ic-cus 192:64 > /tmp/kt_$v.log 2>&1; echo "kt $v rc=$?"
DB=$(find /tmp/kt_$v -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/kernel_stats_$v.md | head -16
python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB > $GRAFT_REPO_ROOT/$O/ba_gaps_$v.txt 2>&1; tail -16 $GRAFT_REPO_ROOT/$O/ba_gaps_$v.txt
