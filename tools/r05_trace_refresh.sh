#!/bin/bash
# step 1 of tools/r05_profiles.sh alone: the headline command under rocprofv3 --kernel-trace with the round's final build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHORT="--no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg"
rocprofv3 --kernel-trace -d $O/trace -o headline -- python $R/bench.py $SHORT > $O/headline_traced_bench_line.json 2> $O/headline_trace.err
python $R/tools/rocpd_summary.py kernels $O/trace/headline_results.db --last-frames 300 > $O/headline_bench_kernel_stats.md
rm -rf $O/trace
head -16 $O/headline_bench_kernel_stats.md; grep -n "feat_ref" $O/headline_bench_kernel_stats.md
