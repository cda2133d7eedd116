#!/bin/bash
mkdir -p gpurun_out/r03_19
run() { n=$1; shift
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop "$@" > gpurun_out/r03_19/$n.json 2> gpurun_out/r03_19/$n.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_19/$n.json').read().strip().splitlines()[-1]); c=d['config']
    print('$n', round(d['value'],1), 'host ms/step', round(c['host_enqueue_ms_per_step'],3))
except Exception as e:
    print('$n FAILED', e); print(open('gpurun_out/r03_19/$n.err').read()[-600:])
PY
}
for cams in 4 0; do
run both_$cams --klt-cams-per-launch $cams
run nosolve_$cams --klt-cams-per-launch $cams --key-every 0
run joint_$cams --klt-cams-per-launch $cams --only-solve joint
run intercam_$cams --klt-cams-per-launch $cams --only-solve intercam
run nosolve_noncc_$cams --klt-cams-per-launch $cams --key-every 0 --no-ncc
run nosolve_nopose_$cams --klt-cams-per-launch $cams --key-every 0 --no-pose
run both_noncc_$cams --klt-cams-per-launch $cams --no-ncc
run both_noposeupd_$cams --klt-cams-per-launch $cams --no-pose-update
done
