#!/bin/bash
# the closing build run long: the C++ frame loop for 6000 frames (twice: digests equal), the Python loop's closed-loop run for 3000
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/soak
mkdir -p $O
cd $R
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
sc = bench.build_scene()
bench.export_workload("/tmp/workload.bin", sc, frames, bench.build_joint_problem(sc), bench.build_ic_problem(sc), 0)
PY
export HSA_KERNARG_POOL_SIZE=$((64 << 20))
for rep in 1 2; do
  timeout 300 tools/cxx/frame_loop.bin /tmp/workload.bin 6000 30 0 2 2> $O/cxx_$rep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cxx loop 6000 frames:', {k:d[k] for k in ('frames_per_s','pose_ok','bmerge_frames','map_points_in_use','second_visit_features_attached','second_visit_conflicts_in_timed_region','second_visit_points_beyond_the_list','register_decisions_unsettled','apply_wait_errors','digest')})" | tee -a $O/summary.txt
done
timeout 600 python tools/r05_drift.py --variant full --frames 3000 --count-attach --out $O/full_3000.jsonl > $O/full_3000.log 2>&1; echo "python loop rc $?" | tee -a $O/summary.txt
tail -1 $O/full_3000.jsonl | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('python loop frame', r['frame'], 'raw', round(r['t_err_max'],3), 'after Sim(3)', round(r['centre_err_sim_max'],4), 'scale', round(r['gauge_sim']['scale'],3))" | tee -a $O/summary.txt
