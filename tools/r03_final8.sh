#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final8; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; grep -c "\[smoke\]" $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver cmd rc=$?"
python -c "
import json
d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1]); c=d['config']; print(round(d['value'],1), 'cxx', c['cxx_frame_loop']['frames_per_s'], 'upload', round(c['with_upload']['frames_per_s'],1), 'frac', round(d['roofline']['frac'],4), 'cpu', round(d['cpu_baseline']['value'],2))"
