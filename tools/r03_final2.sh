#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final2; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; grep smoke $O/smoke.txt | tail -6
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver cmd rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_final2/bench_driver.json").read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
print(round(d["value"],1), "cxx", c["cxx_frame_loop"].get("frames_per_s"), "upload", c["with_upload"]["frames_per_s"], "frac", round(r["frac"],4), "valu", round(r["valu"]["frac"],3), "cfg5klt", c["secondary_cfg5_klt"]["frames_per_s"])
PY
