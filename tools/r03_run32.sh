#!/bin/bash
O=gpurun_out/r03_32; mkdir -p $O
timeout 600 python -m pytest tests/test_ncc_gpu.py -x -q 2>&1 | tail -2
run() { n=$1; shift
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg "$@" > $O/$n.json 2> $O/$n.err
python - <<PY
import json
try:
    d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); c=d['config']; k=c['key_frame_solves_duty']
    print('$n', round(d['value'],1), 'cxx', c['cxx_frame_loop'].get('frames_per_s'), 'joint ms/solve', round(k['joint_ba']['ms_total']/max(k['joint_ba']['solves'],1),3), 'duty', round(k['joint_ba']['share_of_timed_region'],3), c['ncc_matching']['pairs_kept_last_run'])
except Exception as e:
    print('$n FAILED', e); print(open('$O/$n.err').read()[-600:])
PY
}
for rep in 1 2 3; do
run dense_$rep --ncc-dense
run sparse_$rep
done
