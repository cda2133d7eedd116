#!/bin/bash
# the persistent tracker's co-residency: the capacity matrix (tools/r05_diag.py), the KLT parity tests, a headline bench line
mkdir -p gpurun_out/r05d
o=gpurun_out/r05d
timeout 600 python tools/r05_diag.py 2>&1 | grep -v amdgpu.ids | tee $o/diag.txt
python -m pytest tests/test_klt_gpu.py tests/test_configs_gpu.py -m gpu -q > $o/pytest.log 2>&1; tail -2 $o/pytest.log
python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>$o/err.log | tail -1 > $o/bench.json
python - <<PY
import json
j=json.load(open("$o/bench.json")); c=j["config"]
print(round(j["value"],1), "cfg5", {k:c["secondary_cfg5_klt"].get(k) for k in ("frames_per_s","tracker_us_per_frame","tracker_launches_per_frame","ms_per_frame")})
PY
