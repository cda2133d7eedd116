#!/bin/bash
mkdir -p gpurun_out/r03_13
for rep in 1 2; do
for cams in 0 2 3 4; do
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --klt-cams-per-launch $cams > gpurun_out/r03_13/b_${cams}_$rep.json 2> gpurun_out/r03_13/b_${cams}_$rep.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_13/b_${cams}_$rep.json').read().strip().splitlines()[-1])
    print('cams', $cams, 'rep', $rep, round(d['value'],1), d['roofline']['avg_launch_us'], d['roofline']['launches_per_frame'])
except Exception as e:
    print('$cams FAILED', e); print(open('gpurun_out/r03_13/b_${cams}_$rep.err').read()[-800:])
PY
done
done
