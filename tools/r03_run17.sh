#!/bin/bash
mkdir -p gpurun_out/r03_17
for rep in 1 2; do
for nw in 0 1; do
if [ $nw = 1 ]; then export COSLAM_WIN_NOWAIT=1; else unset COSLAM_WIN_NOWAIT; fi
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop > gpurun_out/r03_17/b_${rep}_$nw.json 2> gpurun_out/r03_17/b_${rep}_$nw.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_17/b_${rep}_$nw.json').read().strip().splitlines()[-1]); c=d['config']
    print('rep', $rep, 'nowait', $nw, round(d['value'],1), 'host ms/step', round(c['host_enqueue_ms_per_step'],3), 'max', round(c['host_enqueue_ms_max_step'],3))
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/r03_17/b_${rep}_$nw.err').read()[-800:])
PY
done
done
