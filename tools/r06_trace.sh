#!/bin/bash
# the headline command under rocprofv3 --kernel-trace: per-kernel table over the last 300 frames + the kernels of a few frames per stream
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06t}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHORT="--no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg"
rocprofv3 --kernel-trace -d $O/trace -o headline -- python $R/bench.py $SHORT > $O/headline_traced_bench_line.json 2> $O/headline_trace.err
python $R/tools/rocpd_summary.py kernels $O/trace/headline_results.db --last-frames 300 > $O/headline_bench_kernel_stats.md
python $R/tools/r06_frames.py $O/trace/headline_results.db 120 6 > $O/frames.txt 2>&1
rm -rf $O/trace
head -40 $O/headline_bench_kernel_stats.md
