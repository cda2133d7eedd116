#!/bin/bash
O=gpurun_out/r03_34; mkdir -p $O
./oracle/_ref/ref_ba_dropin_test 2>&1 | tail -2
timeout 600 python -m pytest tests/test_poseupdate_gpu.py -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "pose update"
for rep in 1 2; do
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop > $O/b_$rep.json 2> $O/b_$rep.err
python -c "
import json
d=json.loads(open('$O/b_$rep.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['config']['pose_update']['features_dynamic_last_frame'])"
done
cd /tmp
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 60 --warmup 10 > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats.md | grep -E "pose_update|mergab|ncc_epi"
