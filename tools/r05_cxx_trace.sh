#!/bin/bash
# rocprofv3 kernel trace of the C++ frame loop (tools/cxx/frame_loop.bin) next to the Python loop's: why is it slower?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
python - <<PY
import sys; sys.path.insert(0, "$R")
import bench
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
sc = bench.build_scene()
bench.export_workload("/tmp/workload.bin", sc, frames, bench.build_joint_problem(sc), bench.build_ic_problem(sc), 0)
PY
export HSA_KERNARG_POOL_SIZE=$((64<<20))
tools/cxx/frame_loop.bin /tmp/workload.bin 300 30 0 2 > $O/cxx_line.json 2> $O/cxx.err; cat $O/cxx_line.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o t -- $R/tools/cxx/frame_loop.bin /tmp/workload.bin 300 30 0 2 > $O/cxx_traced.json 2> $O/trace.err
python $R/tools/rocpd_summary.py kernels $O/trace/t_results.db --last-frames 300 > $O/cxx_kernel_stats.md
rm -rf $O/trace
head -40 $O/cxx_kernel_stats.md
sed -n '/per stream/,$p' $O/cxx_kernel_stats.md | head -12
