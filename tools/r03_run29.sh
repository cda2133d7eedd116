#!/bin/bash
O=gpurun_out/r03_29; mkdir -p $O
timeout 600 python -m pytest tests/test_poseupdate_gpu.py -x -q 2>&1 | tail -2
run() { n=$1; shift
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop "$@" > $O/$n.json 2> $O/$n.err
python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); c=d['config']; k=c['key_frame_solves_duty']
    print('$n', round(d['value'],1), 'joint ms/solve', round(k['joint_ba']['ms_total']/max(k['joint_ba']['solves'],1),3))
except Exception as e:
    print('$n FAILED', e); print(open('$O/$n.err').read()[-600:])
PY
}
for rep in 1 2; do
run with_$rep
run without_$rep --no-mergability
done
cd /tmp
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 60 --warmup 10 > /tmp/kt.log 2>&1; echo "kt rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats.md | grep -E "mergab|pose_update|register|intracam|handback" 
