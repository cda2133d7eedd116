# round 3, sweep 3: packed LM-step kernels: parity, then the loop with / without them
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s3; mkdir -p $O
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_posegraph_gpu.py tests/test_sliced_ba_gpu.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -25 | tee $O/pytest.txt
run() { # name, args...
  n=$1; shift
  python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 "$@" > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j['roofline']; c=j['config']
    print(f"{sys.argv[2]:28s} {j['value']:8.1f} frames/s  tracker {r['avg_launch_us']:.1f} us x {r['launches_per_frame']} launches  joint steps {c['joint_ba_last']['lm_steps']} cost {c['joint_ba_last']['cost']:.3f} ic {c['intercam_last']['lm_steps']} cost {c['intercam_last']['cost']:.3f}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
    print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
COSLAM_BA_PACKED=0 run unpacked
run packed
run packed_cams4 --klt-cams-per-launch 4
run packed_k192_ba64 --klt-cus 192 --ba-cus 192:64 --ic-cus 192:64
run packed_k192_ba48_ic16 --klt-cus 192 --ba-cus 192:48 --ic-cus 240:16
run packed_k208_ba48 --klt-cus 208 --ba-cus 208:48 --ic-cus 208:48
run packed_cams4_ba64 --klt-cams-per-launch 4 --ba-cus 192:64 --ic-cus 192:64
cd /tmp
for v in packed packed_k192_ba64; do
  a=""; [ $v = packed_k192_ba64 ] && a="--klt-cus 192 --ba-cus 192:64 --ic-cus 192:64"
  rm -rf /tmp/kt_$v && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 $a > /tmp/kt_$v.log 2>&1; echo "kt $v rc=$?"
  DB=$(find /tmp/kt_$v -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/kernel_stats_$v.md | head -20
  python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB > $GRAFT_REPO_ROOT/$O/ba_gaps_$v.txt 2>&1; tail -18 $GRAFT_REPO_ROOT/$O/ba_gaps_$v.txt
done
