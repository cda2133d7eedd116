# round 3, sweep 1: how the frame rate depends on how the persistent tracker shares the chip with the key-frame solves
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s1; mkdir -p $O
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
run cams4 --klt-cams-per-launch 4
run cams2 --klt-cams-per-launch 2
run cams6 --klt-cams-per-launch 6
run mask192 --klt-cus 192
run mask128 --klt-cus 128
run base_nosolve --key-every 0
run cams4_nosolve --klt-cams-per-launch 4 --key-every 0
run base_noreg --no-register
run cams4_noreg --klt-cams-per-launch 4 --no-register
run cams4_joint --klt-cams-per-launch 4 --only-solve joint
run base_driver --steps 20 --warmup 5
run cams4_driver --klt-cams-per-launch 4 --steps 20 --warmup 5
cd /tmp
for v in base cams4; do
  a=""; [ $v = cams4 ] && a="--klt-cams-per-launch 4"
  rm -rf /tmp/kt_$v && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 $a > /tmp/kt_$v.log 2>&1; echo "kt $v rc=$?"
  DB=$(find /tmp/kt_$v -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/kernel_stats_$v.md | head -24
  python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB > $GRAFT_REPO_ROOT/$O/ba_gaps_$v.txt 2>&1; tail -25 $GRAFT_REPO_ROOT/$O/ba_gaps_$v.txt
  python $GRAFT_REPO_ROOT/tools/timeline.py $DB 2500 0.5 > $GRAFT_REPO_ROOT/$O/timeline_$v.txt 2>&1
done
