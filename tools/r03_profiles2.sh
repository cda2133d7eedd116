#!/bin/bash
# refresh of the headline trace for the round's closing code (all cameras in one tracker launch, mergability in the loop)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p2; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/kt && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg > $O/headline_traced_bench.json 2> /tmp/kt.err; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $O/headline_kernel_stats.md | head -14
python $R/tools/ba_gaps.py $DB > $O/ba_in_loop_durations_and_gaps.txt 2>&1
python $R/tools/key_interval.py $DB > $O/key_interval.txt 2>&1
python -c "
import json
d=json.loads(open('$O/headline_traced_bench.json').read().strip().splitlines()[-1]); print('traced run:', round(d['value'],1), 'roofline avg_launch_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
