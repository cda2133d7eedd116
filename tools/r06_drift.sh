#!/bin/bash
# the closed loop against the synthetic truth with round 6's registration (second visits, fused launches) and the classification over references:
# 1500 frames, the full variant, against the same loop with the classification back on this frame's features and with the rounds off
R=${GRAFT_REPO_ROOT:-$(pwd)}
o=$R/gpurun_out/r06/drift
mkdir -p $o
cd $R
for v in full classify_plain no_rounds; do
  timeout 300 python tools/r05_drift.py --variant $v --frames 1500 --count-attach --out $o/$v.jsonl > $o/$v.log 2>&1 || echo "drift $v rc=$?"
done
python tools/r05_drift_summary.py $o 500 1000 1500 > $o/summary.md 2>&1; cat $o/summary.md | cut -c1-400
