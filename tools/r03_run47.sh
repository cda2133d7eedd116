#!/bin/bash
O=gpurun_out/r03_47; mkdir -p $O
COSLAM_BA_SEGTIME=1 timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop > $O/b.json 2> $O/b.err
grep "ba segtime" $O/b.err | tail -24
python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; j=c['key_frame_solves_duty']['joint_ba']; print(round(d['value'],1), 'ms/solve', round(j['ms_total']/j['solves'],3), 'parse', round(j['ms_parse_total']/j['solves'],3))"
