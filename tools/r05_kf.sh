#!/bin/bash
mkdir -p gpurun_out/r05k
o=gpurun_out/r05k
python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>$o/err.log | tail -1 > $o/bench.json
python - <<PY
import json
j=json.load(open("$o/bench.json")); c=j["config"]
print(round(j["value"],1), json.dumps(c.get("secondary_keyframe_decision"))[:1500])
print("refs", json.dumps(c.get("feature_references"))[:300])
PY
tail -3 $o/err.log
python -m pytest tests/test_bench_contract_gpu.py -m gpu -q -x > $o/pytest.log 2>&1; tail -3 $o/pytest.log
