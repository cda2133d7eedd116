#!/bin/bash
mkdir -p gpurun_out/r05c2
o=gpurun_out/r05c2
python -m pytest tests/test_poseupdate_gpu.py -m gpu -q -x -k "feature_references or relinked" > $o/pytest2.log 2>&1; tail -2 $o/pytest2.log
for x in 1 0 1 0; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --feature-chains $x 2>$o/err_$x.log | tail -1 > $o/bench_chains$x.json
  python - <<PY
import json
try:
    j=json.load(open("$o/bench_chains$x.json")); c=j["config"]
    print("chains=$x", round(j["value"],1), "rig", c.get("rig_error_vs_truth",{}).get("centres_after_sim3_max"))
except Exception as e:
    print("chains=$x FAILED", e)
PY
done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
rocprofv3 --kernel-trace -d $R/$o/trace -o t -- python $R/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg > $R/$o/traced_line.json 2> $R/$o/trace.err
python $R/tools/rocpd_summary.py kernels $R/$o/trace/t_results.db --last-frames 300 > $R/$o/kernel_stats.md
rm -rf $R/$o/trace
head -40 $R/$o/kernel_stats.md | cut -c1-200
