#!/bin/bash
O=gpurun_out/r03_42; mkdir -p $O
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_cxx_dropin_gpu.py -x -q -k "window or dropin or parse" 2>&1 | tail -5
for rep in 1 2; do
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg > $O/b_$rep.json 2> $O/b.err || tail -5 $O/b.err
python -c "
import json
d=json.loads(open('$O/b_$rep.json').read().strip().splitlines()[-1]); c=d['config']; j=c['key_frame_solves_duty']['joint_ba']; print(round(d['value'],1), 'duty', round(j['share_of_timed_region'],3), 'parse ms/solve', round(j['ms_parse_total']/j['solves'],4), 'ms/solve', round(j['ms_total']/j['solves'],3), (c.get('cxx_frame_loop') or {}).get('frames_per_s'))"
done
