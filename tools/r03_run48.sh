#!/bin/bash
O=gpurun_out/r03_48; mkdir -p $O
for rep in 1 2; do
for v in base prio2 prio3; do
if [ $v = base ]; then unset COSLAM_HIP_LIB; else export COSLAM_HIP_LIB=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip_$v.so; fi
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop > $O/b_${v}_$rep.json 2> $O/b.err || tail -5 $O/b.err
python -c "
import json
d=json.loads(open('$O/b_${v}_$rep.json').read().strip().splitlines()[-1]); c=d['config']; j=c['key_frame_solves_duty']['joint_ba']; k=c['key_frame_solves_duty']['inter_camera']; print('$v', round(d['value'],1), 'duty', round(j['share_of_timed_region'],3), 'ms/solve', round(j['ms_total']/j['solves'],3), 'ic', round(k['ms_total']/k['solves'],3), 'trk us', round(d['roofline']['avg_launch_us'],1))"
done
done
