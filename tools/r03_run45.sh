#!/bin/bash
O=gpurun_out/r03_45; mkdir -p $O
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_posegraph_gpu.py tests/test_cxx_dropin_gpu.py -x -q 2>&1 | tail -4
for rep in 1 2; do
for v in 0 1; do
COSLAM_BA_SPLIT_TR=$v timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop > $O/b_${v}_$rep.json 2> $O/b.err || tail -5 $O/b.err
python -c "
import json
d=json.loads(open('$O/b_${v}_$rep.json').read().strip().splitlines()[-1]); c=d['config']; j=c['key_frame_solves_duty']['joint_ba']; k=c['key_frame_solves_duty']['inter_camera']; print('split=$v', round(d['value'],1), 'duty', round(j['share_of_timed_region'],3), 'ms/solve', round(j['ms_total']/j['solves'],3), 'ic ms/solve', round(k['ms_total']/k['solves'],3), 'lm', c['joint_ba_last']['lm_steps'], c['intercam_last']['lm_steps'], c['intercam_last']['cost'])"
done
done
