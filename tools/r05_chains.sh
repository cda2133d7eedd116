#!/bin/bash
# feature references (stale features as views, re-linked chains) in the closed loop: parity tests, drift with / without, bench A/B
mkdir -p gpurun_out/r05c2
o=gpurun_out/r05c2
python -m pytest tests/test_poseupdate_gpu.py tests/test_bench_contract_gpu.py -m gpu -q -x > $o/pytest.log 2>&1; tail -3 $o/pytest.log
for v in full no_chains; do
  timeout 250 python tools/r05_drift.py --variant $v --frames 1500 --count-attach --out $o/$v.jsonl > $o/$v.log 2>&1 || echo "drift $v rc=$?"
done
python tools/r05_drift_summary.py $o 500 1000 1500 > $o/summary.md 2>&1; cat $o/summary.md | cut -c1-400
for x in 1 0 1 0; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --feature-chains $x 2>$o/err_$x.log | tail -1 > $o/bench_chains$x.json
  python - <<PY
import json
try:
    j=json.load(open("$o/bench_chains$x.json")); c=j["config"]
    print("chains=$x", round(j["value"],1), "rig", c.get("rig_error_vs_truth",{}).get("centres_after_sim3_max"), "refs", c.get("feature_references"))
except Exception as e:
    print("chains=$x FAILED", e)
PY
done
