#!/bin/bash
# diagnostic: the C++ loop with the decision placing the key frames (ratio 1.5: a key frame per frame), one rank against two ranks on one GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
[ -f /tmp/workload.bin ] || python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
sc = bench.build_scene()
bench.export_workload("/tmp/workload.bin", sc, frames, bench.build_joint_problem(sc), bench.build_ic_problem(sc), 0)
PY
export HSA_KERNARG_POOL_SIZE=$((64 << 20)) COSLAM_KLT_FUSED=0 COSLAM_KEYFRAME_DRIVES=1 COSLAM_KEYFRAME_LAG=1 COSLAM_KEYFRAME_RATIO=1.5
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d["key_frames_placed_by_the_decision"]; print("rank", d["rank"], "of", d["world"], ":", len(p), "placed; first gap at", next((i+1 for i,(a,b) in enumerate(zip(p,p[1:])) if b!=a+1), None), "applied", d["windows_applied"], "requested", d["windows_requested"], "wait errors", d["apply_wait_errors"], "digest", d["digest"])'
tools/cxx/frame_loop.bin /tmp/workload.bin 60 10 0 2 2>/dev/null | python -c "$show"
for rep in 1 2 3 4; do
  for r in 0 1; do
    RANK=$r WORLD_SIZE=2 COSLAM_FORCE_DEVICE=0 COSLAM_COMM=host:/kf2_$$_$rep tools/cxx/frame_loop.bin /tmp/workload.bin 60 10 0 2 2>/tmp/err_$r.txt | python -c "$show" &
  done
  wait
done
