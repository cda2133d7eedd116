#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/merge_print
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
COSLAM_MERGE_PRINT=1 $R/tools/cxx/frame_loop.bin /tmp/workload.bin 300 30 0 2 > $O/out.txt 2> $O/err.txt
grep "k_decide_merge\|k_revisit_decide" $O/out.txt | sed -n "1,400p" | awk "NR%9==0" | head -60
tail -c 400 $O/out.txt
