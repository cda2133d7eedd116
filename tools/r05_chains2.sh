#!/bin/bash
mkdir -p gpurun_out/r05c3
o=gpurun_out/r05c3
(time python -m pytest tests -m gpu -q) > $o/pytest_gpu.log 2>&1; tail -4 $o/pytest_gpu.log
for x in 1 0 1; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --feature-chains $x 2>$o/err_$x.log | tail -1 > $o/bench_chains$x.json
  python - <<PY
import json
try:
    j=json.load(open("$o/bench_chains$x.json")); c=j["config"]
    print("chains=$x", round(j["value"],1), "rig", c.get("rig_error_vs_truth",{}).get("centres_after_sim3_max"))
except Exception as e:
    print("chains=$x FAILED", e)
PY
done
