#!/bin/bash
# A/B on one box, alternating: the persistent tracker's placement (cs_klt_set_xcd_placement) x the LDS patch margin
# (CS_ROWS_MARGIN=1: 13.25 KB of LDS per wave: 52 KB of a CU's LDS free beside its two tracker workgroups instead of 32)
mkdir -p gpurun_out/r05x
o=gpurun_out/r05x
M1=$PWD/coslam_amd/lib/libcoslam_hip_margin1.so
COSLAM_HIP_LIB=$M1 python -m pytest tests/test_klt_gpu.py -m gpu -q > $o/pytest_klt_margin1.log 2>&1; tail -2 $o/pytest_klt_margin1.log
run() {  # name lib xcd
  COSLAM_HIP_LIB=$2 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --klt-xcd-placement $3 2>$o/err_$1.log | tail -1 > $o/bench_$1.json
  python - <<PY
import json
try:
    j=json.load(open("$o/bench_$1.json"))
    r=j["roofline"]
    print("$1", round(j["value"],1), "ms", round(j["ms_per_step"],4), "tracker us alone", round(r.get("avg_launch_us"),1), "in loop", round(r.get("avg_launch_us_in_loop"),1), "frac", round(r.get("frac"),4))
except Exception as e:
    print("$1 FAILED", e)
PY
}
for rep in 1 2; do
  run base_$rep "" 0
  run m1_xcd0_$rep $M1 0
  run m1_xcd1_$rep $M1 1
done
