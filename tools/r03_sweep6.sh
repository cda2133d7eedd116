# round 3, sweep 6: how many hardware queues the runtime may use (streams beyond that share one: head-of-line blocking)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s6; mkdir -p $O
run() { # name, args...
  n=$1; shift
  timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 "$@" > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} {j['value']:8.1f} frames/s")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
export COSLAM_BA_PACKED=0
for rep in 1 2; do
run q_default_$rep
GPU_MAX_HW_QUEUES=2 run q2_$rep
GPU_MAX_HW_QUEUES=8 run q8_$rep
GPU_MAX_HW_QUEUES=16 run q16_$rep
GPU_MAX_HW_QUEUES=8 run q8_regstream_$rep --reg-stream 1
GPU_MAX_HW_QUEUES=8 run q8_cams4_$rep --klt-cams-per-launch 4
GPU_MAX_HW_QUEUES=8 COSLAM_BA_PACKED=1 run q8_packed_$rep
done
cd /tmp
export GPU_MAX_HW_QUEUES=8
rm -rf /tmp/kt_q8 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_q8 -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > /tmp/kt_q8.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt_q8 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB > $GRAFT_REPO_ROOT/$O/ba_gaps_q8.txt 2>&1; tail -16 $GRAFT_REPO_ROOT/$O/ba_gaps_q8.txt
