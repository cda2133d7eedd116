#!/bin/bash
O=gpurun_out/r03_37; mkdir -p $O
timeout 900 python -m pytest tests/test_pose_ba_gpu.py -x -q -k "packed" 2>&1 | tail -3
run() { n=$1; shift
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop "$@" > $O/$n.json 2> $O/$n.err
python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); c=d['config']; k=c['key_frame_solves_duty']
    print('$n', round(d['value'],1), 'joint ms/solve', round(k['joint_ba']['ms_total']/max(k['joint_ba']['solves'],1),3), 'ic ms/solve', round(k['inter_camera']['ms_total']/max(k['inter_camera']['solves'],1),3), c['intercam_last'])
except Exception as e:
    print('$n FAILED', e); print(open('$O/$n.err').read()[-600:])
PY
}
for rep in 1 2 3; do
run base_$rep
run ic1_$rep --ba-persist 0:1
done
