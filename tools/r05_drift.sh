#!/bin/bash
# The drift campaign of round 5 (run on the GPU box: gpurun -- tools/r05_drift.sh): tools/r05_drift.py for every variant, one process
# each, 1500 frames, into gpurun_out/r05/drift_final/; profiles/r05_drift_final/ + tools/r05_drift_summary.py make the table of it.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/drift_final
mkdir -p $O
cd $R
for v in full r03_like no_writeback no_decide no_ncc lag1 lag4 points_only poses_only no_false no_update no_classify no_merge; do
  timeout 200 python tools/r05_drift.py --variant $v --frames 1500 --count-attach --out $O/$v.jsonl > $O/$v.log 2>&1 || echo "drift $v rc=$?"
done
timeout 200 python tools/r05_drift.py --variant full --frames 1500 --count-attach --time-intracam --pixel-err-reading std --out $O/full_std_reading.jsonl > $O/full_std_reading.log 2>&1
timeout 200 python tools/r05_drift.py --variant full --frames 1500 --time-intracam --out $O/full_intracam_alone.jsonl > $O/full_intracam_alone.log 2>&1
ls -la $O
