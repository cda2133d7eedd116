#!/bin/bash
O=gpurun_out/r03_39; mkdir -p $O
timeout 600 python -m pytest tests/test_poseupdate_gpu.py -x -q 2>&1 | tail -15
for v in "" "--no-update-points"; do
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg $v > $O/b.json 2> $O/b.err || tail -5 $O/b.err
python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; print('$v', round(d['value'],1), c.get('update_new_poses_points'), (c.get('cxx_frame_loop') or {}).get('frames_per_s'), (c.get('cxx_frame_loop') or {}).get('update_new_poses_points_runs'))"
done
