#!/bin/bash
mkdir -p gpurun_out/r03_15
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_cxx_dropin_gpu.py tests/test_bench_contract_gpu.py -x -q > gpurun_out/r03_15/pytest.txt 2>&1; tail -5 gpurun_out/r03_15/pytest.txt
for ch in 2 3 5; do
for cams in 4 0; do
COSLAM_BA_WINDOW_CHUNK=$ch timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --klt-cams-per-launch $cams > gpurun_out/r03_15/b_${ch}_$cams.json 2> gpurun_out/r03_15/b_${ch}_$cams.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_15/b_${ch}_$cams.json').read().strip().splitlines()[-1]); c=d['config']
    print('chunk', $ch, 'cams', $cams, round(d['value'],1), 'cxx', c['cxx_frame_loop'].get('frames_per_s'), c['joint_ba_last'])
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/r03_15/b_${ch}_$cams.err').read()[-800:])
PY
done
done
