#!/bin/bash
# pose update in the frame loop: bench driver command with / without it, kernel trace of the headline
mkdir -p gpurun_out/r03_11
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" "--no-pose-update"; do
  n=$( [ -z "$v" ] && echo with || echo without )
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $v > gpurun_out/r03_11/bench_$n.json 2> gpurun_out/r03_11/bench_$n.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_11/bench_$n.json').read().strip().splitlines()[-1])
    c=d['config']; print('$n', round(d['value'],1), 'cxx', c['cxx_frame_loop'].get('frames_per_s'), 'upload', c['with_upload']['frames_per_s'] if c.get('with_upload') else None)
    print('   pose_update', c.get('pose_update')); print('   pose_err', c['pose_translation_error_vs_truth'], 'npts', c['pose_correspondences'])
    print('   joint', c['joint_ba_last'], c['joint_ba_problem'].get('points'))
except Exception as e:
    print('$n FAILED', e); print(open('gpurun_out/r03_11/bench_$n.err').read()[-1500:])
PY
done
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r03_11/bench_300.json 2> gpurun_out/r03_11/bench_300.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_11/bench_300.json').read().strip().splitlines()[-1]); c=d['config']
print('300 steps', round(d['value'],1), c['pose_update'], c['pose_translation_error_vs_truth'])"
cd /tmp
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 100 --warmup 10 > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/gpurun_out/r03_11/kernel_stats.md | head -24
python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB > $GRAFT_REPO_ROOT/gpurun_out/r03_11/ba_gaps.txt 2>&1; tail -12 $GRAFT_REPO_ROOT/gpurun_out/r03_11/ba_gaps.txt
python $GRAFT_REPO_ROOT/tools/timeline.py $DB 2>/dev/null | head -0
