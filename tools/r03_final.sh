#!/bin/bash
# round-3 closing run: full GPU suite, smoke, the driver's bench command (x3), the default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -6 $O/smoke.txt
for k in 1 2 3; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_$k.json 2> $O/bench_driver_$k.err; echo "driver cmd $k rc=$?"
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout 900 python bench.py --klt-cams-per-launch 4 --no-cpu-baseline --no-secondary > $O/bench_default_cams4.json 2> $O/bench_default_cams4.err; echo "cams4 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_final/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
        print(f.split("/")[-1], round(d["value"],1), "cxx", (c.get("cxx_frame_loop") or {}).get("frames_per_s"), "upload", (c.get("with_upload") or {}).get("frames_per_s"),
              "frac", round(r["frac"],4), "valu", None if not r.get("valu") else round(r["valu"]["frac"],3), "cpu", (d.get("cpu_baseline") or {}).get("value"),
              "cfg2", (c.get("secondary_cfg2") or {}).get("camera_frames_per_s"), "cfg5klt", (c.get("secondary_cfg5_klt") or {}).get("frames_per_s"))
    except Exception as e:
        print(f, "FAILED", e)
PY
