# round 3, sweep 4: packed (per-phase launches) vs persistent (one launch per LM run): parity, then the loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s4; mkdir -p $O
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_posegraph_gpu.py -x -q -m gpu 2>&1 | tail -25 | tee $O/pytest.txt
run() { # name, args...
  n=$1; shift
  timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 "$@" > $O/$n.json 2> $O/$n.err
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
run persist48_8 --ba-persist 48:8
run persist48_8_cams4 --ba-persist 48:8 --klt-cams-per-launch 4
run persist64_8_cams4 --ba-persist 64:8 --klt-cams-per-launch 4
run persist32_8_cams4 --ba-persist 32:8 --klt-cams-per-launch 4
run persist40_4_cams4 --ba-persist 40:4 --klt-cams-per-launch 4
COSLAM_BA_PERSIST_LDS_KB=40 run persist48_8_cams4_lds40 --ba-persist 48:8 --klt-cams-per-launch 4
cd /tmp
for v in persist48_8_cams4; do
  rm -rf /tmp/kt_$v && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --ba-persist 48:8 --klt-cams-per-launch 4 > /tmp/kt_$v.log 2>&1; echo "kt $v rc=$?"
  DB=$(find /tmp/kt_$v -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/kernel_stats_$v.md | head -24
  python $GRAFT_REPO_ROOT/tools/timeline.py $DB 3000 0.5 > $GRAFT_REPO_ROOT/$O/timeline_$v.txt 2>&1
done
