#!/bin/bash
mkdir -p gpurun_out/r03_14
cd /tmp
for cams in 4 0; do
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 100 --warmup 10 --klt-cams-per-launch $cams > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/key_interval.py $DB > $GRAFT_REPO_ROOT/gpurun_out/r03_14/key_interval_cams$cams.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/gpurun_out/r03_14/kernel_stats_cams$cams.md > /dev/null
done
