#!/bin/bash
mkdir -p gpurun_out/r03_20
run() { n=$1; shift
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop "$@" > gpurun_out/r03_20/$n.json 2> gpurun_out/r03_20/$n.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_20/$n.json').read().strip().splitlines()[-1]); c=d['config']; k=c['key_frame_solves_duty']
    print('$n', round(d['value'],1), 'joint', k['joint_ba']['solves'], round(k['joint_ba']['ms_total']/max(k['joint_ba']['solves'],1),3), 'ms/solve share', round(k['joint_ba']['share_of_timed_region'],3), 'parse ms/solve', round(k['joint_ba']['ms_parse_total']/max(k['joint_ba']['solves'],1),3), 'lm', c['joint_ba_last'], 'ic', round(k['inter_camera']['ms_total']/max(k['inter_camera']['solves'],1),3), 'share', round(k['inter_camera']['share_of_timed_region'],3))
except Exception as e:
    print('$n FAILED', e); print(open('gpurun_out/r03_20/$n.err').read()[-600:])
PY
}
for cams in 4 0; do
run both_$cams --klt-cams-per-launch $cams
run joint_$cams --klt-cams-per-launch $cams --only-solve joint
run both_noncc_$cams --klt-cams-per-launch $cams --no-ncc
done
