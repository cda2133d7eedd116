#!/bin/bash
# which m_mappedPtsReduceRatio places how many key frames in the C++ loop's 120 frames (tests/test_cxx_dropin_gpu.py picks a sparse one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
sc = bench.build_scene()
bench.export_workload("/tmp/workload.bin", sc, frames, bench.build_joint_problem(sc), bench.build_ic_problem(sc), 0)
PY
export HSA_KERNARG_POOL_SIZE=$((64 << 20)) COSLAM_KLT_FUSED=0
for r in 1.25 1.3 1.35 1.4 1.45; do
  COSLAM_KEYFRAME_DRIVES=1 COSLAM_KEYFRAME_LAG=1 COSLAM_KEYFRAME_RATIO=$r tools/cxx/frame_loop.bin /tmp/workload.bin 60 10 0 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['key_frames_placed_by_the_decision']; print('ratio $r:', len(p), 'key frames', p[:30], 'applied', d['windows_applied'], 'wait errors', d['apply_wait_errors'])"
done
