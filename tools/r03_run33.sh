#!/bin/bash
O=gpurun_out/r03_33; mkdir -p $O
timeout 900 python -m pytest tests/test_klt_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -3
python tools/group_cam.py quick 2>&1 | grep -v "^/opt" | head -6
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop > $O/b.json 2> $O/b.err
python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['avg_launch_us'], d['roofline']['frac'])"
