#!/bin/bash
mkdir -p gpurun_out/r03_21
run() { n=$1; shift
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop "$@" > gpurun_out/r03_21/$n.json 2> gpurun_out/r03_21/$n.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_21/$n.json').read().strip().splitlines()[-1]); c=d['config']; k=c['key_frame_solves_duty']
    print('$n', round(d['value'],1), 'joint ms/solve', round(k['joint_ba']['ms_total']/max(k['joint_ba']['solves'],1),3), 'share', round(k['joint_ba']['share_of_timed_region'],3), 'ic ms/solve', round(k['inter_camera']['ms_total']/max(k['inter_camera']['solves'],1),3))
except Exception as e:
    print('$n FAILED', e); print(open('gpurun_out/r03_21/$n.err').read()[-600:])
PY
}
for rep in 1 2; do
for ch in 1 2 3; do
export COSLAM_BA_WINDOW_CHUNK=$ch
run chunk${ch}_$rep
done
done
unset COSLAM_BA_WINDOW_CHUNK
for ch in 1 2 5; do
export COSLAM_BA_CHUNK=$ch
run icchunk${ch}
done
