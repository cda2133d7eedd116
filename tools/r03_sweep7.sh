# round 3, sweep 7: the flow schedule (Schur | solver | update concurrent, ordered by flags)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s7; mkdir -p $O
timeout 900 python -m pytest tests/test_pose_ba_gpu.py -x -q -m gpu -k "packed or async" 2>&1 | tail -15 | tee $O/pytest.txt
run() { # name, args...
  n=$1; shift
  timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 "$@" > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=j['config']
    print(f"{sys.argv[2]:28s} {j['value']:8.1f} frames/s  joint {c['joint_ba_last']['lm_steps']} {c['joint_ba_last']['cost']:.3f} ic {c['intercam_last']['lm_steps']} {c['intercam_last']['cost']:.3f}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
for rep in 1 2; do
COSLAM_BA_PACKED=0 run unpacked_$rep
run packed_$rep
COSLAM_BA_FLOW=1 run flow_$rep
COSLAM_BA_FLOW=1 run flow_cams4_$rep --klt-cams-per-launch 4
COSLAM_BA_PACKED=0 run unpacked_cams4_$rep --klt-cams-per-launch 4
done
cd /tmp
export COSLAM_BA_FLOW=1
rm -rf /tmp/kt_f && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_f -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > /tmp/kt_f.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt_f -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/kernel_stats_flow.md | head -16
python $GRAFT_REPO_ROOT/tools/timeline.py $DB 1500 0.5 > $GRAFT_REPO_ROOT/$O/timeline_flow.txt 2>&1
grep -E "k_lin_packed|k_schur_flow|k_solve_flow|k_update_flow|k_control_final" $GRAFT_REPO_ROOT/$O/timeline_flow.txt | head -60
