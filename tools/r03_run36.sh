#!/bin/bash
O=gpurun_out/r03_36; mkdir -p $O
timeout 600 python -m pytest tests/test_ncc_gpu.py tests/test_bench_contract_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg > $O/b_$rep.json 2> $O/b_$rep.err
python -c "
import json
d=json.loads(open('$O/b_$rep.json').read().strip().splitlines()[-1]); c=d['config']; print(round(d['value'],1), 'cxx', c['cxx_frame_loop'].get('frames_per_s') or c['cxx_frame_loop'], c['ncc_matching']['pairs_kept_last_run'])" || tail -5 $O/b_$rep.err
done
