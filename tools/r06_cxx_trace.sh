#!/bin/bash
# the C++ frame loop (the bench's `value`) under rocprofv3 --kernel-trace: per-kernel table over the last 300 frames + the kernels of a few
# frames per stream
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06/cxx_trace}
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
cd /tmp && export TMPDIR=/tmp
export HSA_KERNARG_POOL_SIZE=$((64 << 20))
$R/tools/cxx/frame_loop.bin /tmp/workload.bin 300 30 0 2 > $O/untraced_line.json 2> $O/untraced.err
rocprofv3 --kernel-trace -d $O/trace -o cxx -- $R/tools/cxx/frame_loop.bin /tmp/workload.bin 300 30 0 2 > $O/traced_line.json 2> $O/trace.err
python $R/tools/rocpd_summary.py kernels $O/trace/cxx_results.db --last-frames 300 > $O/cxx_loop_kernel_stats.md
python $R/tools/r06_frames.py $O/trace/cxx_results.db 120 8 > $O/frames.txt 2>&1
rm -rf $O/trace
head -50 $O/cxx_loop_kernel_stats.md
cat $O/untraced_line.json | head -c 600
