#!/bin/bash
mkdir -p gpurun_out/r03_18
timeout 600 python -m pytest tests/test_pose_ba_gpu.py -x -q > gpurun_out/r03_18/pytest.txt 2>&1; tail -3 gpurun_out/r03_18/pytest.txt
for rep in 1 2; do
for cams in 4 0; do
for nw in 16 8 4 0; do
if [ $nw = 0 ]; then unset COSLAM_BA_SOLVE_WAVES; else export COSLAM_BA_SOLVE_WAVES=$nw; fi
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop --klt-cams-per-launch $cams > gpurun_out/r03_18/b_${rep}_${cams}_$nw.json 2> gpurun_out/r03_18/b_${rep}_${cams}_$nw.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_18/b_${rep}_${cams}_$nw.json').read().strip().splitlines()[-1]); c=d['config']
    print('rep', $rep, 'cams', $cams, 'solve waves', $nw, round(d['value'],1))
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/r03_18/b_${rep}_${cams}_$nw.err').read()[-800:])
PY
done
done
done
