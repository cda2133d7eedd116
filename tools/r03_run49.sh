#!/bin/bash
O=gpurun_out/r03_49; mkdir -p $O
for v in "" "--no-classify"; do
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg $v > $O/b.json 2> $O/b.err || tail -5 $O/b.err
python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; pu=c['pose_update']; print('$v', round(d['value'],1), pu.get('map_points_classify'), 'uncertain', pu['map_points_uncertain'], 'cxx', (c.get('cxx_frame_loop') or {}).get('frames_per_s'), c['pose_ok'], c['pose_correspondences'])"
done
