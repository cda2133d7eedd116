#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R
( time python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); c = d["config"]
print("default", round(d["value"], 1), "cxx", (c.get("cxx_frame_loop") or {}).get("frames_per_s"), "upload", (c.get("with_upload") or {}).get("frames_per_s"), "cpu", d["cpu_baseline"]["value"])
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k not in ("traffic_source", "valu")}))
print("rig", json.dumps(c["rig_error_vs_truth"])[:400])
PY
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
python - <<PY
import json
d = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1]); print("driver cmd", round(d["value"], 1), d["ms_per_step"])
PY
