#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final5; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; grep -c "\[smoke\]" $O/smoke.txt
for k in 1 2 3; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_$k.json 2> $O/bench_driver_$k.err; echo "driver cmd $k rc=$?"
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_final5/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
        print(f.split("/")[-1], round(d["value"],1), "cxx", (c.get("cxx_frame_loop") or {}).get("frames_per_s"), "upload", (c.get("with_upload") or {}).get("frames_per_s"),
              "frac", round(r["frac"],4), "valu", round(r["valu"]["frac"],3), "cpu", (d.get("cpu_baseline") or {}).get("value"), "cfg5klt", (c.get("secondary_cfg5_klt") or {}).get("frames_per_s"),
              "duty", round(c["key_frame_solves_duty"]["joint_ba"]["share_of_timed_region"],3))
    except Exception as e:
        print(f, "FAILED", e)
PY
cd /tmp
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 100 --warmup 10 > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_final5/kernel_stats.md | head -12
grep '^{"metric"' /tmp/kt.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/r03_final5/traced_bench_line.json
