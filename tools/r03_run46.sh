#!/bin/bash
O=gpurun_out/r03_46; mkdir -p $O
timeout 600 python -m pytest tests/test_poseupdate_gpu.py -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "\[smoke\]"
cd /tmp
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 100 --warmup 10 > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats.md | grep -E "update_points|pose_update|mergab"
grep '^{"metric"' /tmp/kt.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('traced', round(d['value'],1), d['config']['update_new_poses_points'])"
