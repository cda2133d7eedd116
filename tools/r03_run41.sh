#!/bin/bash
O=gpurun_out/r03_41; mkdir -p $O
timeout 600 python -m pytest tests/test_poseupdate_gpu.py -x -q 2>&1 | tail -5
for v in "" "--no-update-points"; do
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg $v > $O/b.json 2> $O/b.err || tail -5 $O/b.err
python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; u=c.get('update_new_poses_points') or {}; print('$v', round(d['value'],1), u.get('runs_in_timed_region'), u.get('static_points_retriangulated_last_run'), u.get('dynamic_points_retriangulated_last_run'), (c.get('cxx_frame_loop') or {}).get('frames_per_s'), (c.get('cxx_frame_loop') or {}).get('update_new_poses_points_runs'))"
done
cd /tmp
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 100 --warmup 10 > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats.md | head -40
tail -1 /tmp/kt.log > $GRAFT_REPO_ROOT/$O/traced_bench_line.json
