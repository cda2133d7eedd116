#!/bin/bash
O=gpurun_out/r03_27; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver cmd rc=$?"
timeout 900 python bench.py --no-cpu-baseline --no-secondary > $O/bench_300.json 2> $O/bench_300.err; echo "300 rc=$?"
python - <<'PY'
import json
for n in ("bench_driver","bench_300"):
    try:
        d=json.loads(open(f"gpurun_out/r03_27/{n}.json").read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
        print(n, round(d["value"],1), "cxx", c["cxx_frame_loop"].get("frames_per_s") or c["cxx_frame_loop"], "upload", c["with_upload"]["frames_per_s"], "frac", round(r["frac"],4), c["register_candidates_last_frame"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "FAILED", e); print(open(f"gpurun_out/r03_27/{n}.err").read()[-800:])
PY
timeout 600 python -m pytest tests/test_bench_contract_gpu.py -x -q 2>&1 | tail -3
