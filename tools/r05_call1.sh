#!/bin/bash
# GPU call 1 of round 5: the GPU test suite, the drift variants (tools/r05_drift.py), scheduling / CU-partition A/B runs of the
# headline loop, and a kernel trace of the base loop and of the best-looking schedule.  Everything lands under gpurun_out/r05/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O/drift $O/ab $O/trace
cd $R
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
for v in full no_writeback r03_like no_decide no_ncc no_decide_no_ncc lag1 lag4 points_only poses_only no_false no_update no_classify no_merge no_intercam; do
  timeout 120 python tools/r05_drift.py --variant $v --frames 1500 --out $O/drift/$v.jsonl > $O/drift/$v.log 2>&1 || echo "drift $v rc=$?"
done
timeout 120 python tools/r05_drift.py --variant full --frames 1500 --hist 256 --out $O/drift/full_hist256.jsonl > $O/drift/full_hist256.log 2>&1
timeout 120 python tools/r05_drift.py --variant full --frames 1500 --min-distance 8 --out $O/drift/full_mindist8.jsonl > $O/drift/full_mindist8.log 2>&1
SHORT="--no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg"
ab() { name=$1; shift; timeout 180 python bench.py $SHORT "$@" > $O/ab/$name.json 2> $O/ab/$name.err || echo "ab $name rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("$O/ab/$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"], 1), "frames/s", "pose_err", round(d["config"]["pose_translation_error_vs_truth"], 4))
except Exception as e:
    print("$name", "FAILED", e)
PY
}
ab base1
ab after_intracam --klt-after-intracam 1
ab klt192_pose64 --klt-cus 192 --pose-cus 64
ab klt192 --klt-cus 192
ab base2
ab klt160_pose96 --klt-cus 160 --pose-cus 96
ab klt128_pose128 --klt-cus 128 --pose-cus 128
ab klt192_pose64_after --klt-cus 192 --pose-cus 64 --klt-after-intracam 1
ab after_intracam2 --klt-after-intracam 1
ab klt224_pose32 --klt-cus 224 --pose-cus 32
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace/base -o t -- python $R/bench.py $SHORT > $O/trace/base_line.json 2> $O/trace/base.err
python $R/tools/rocpd_summary.py kernels $O/trace/base/t_results.db --last-frames 300 > $O/trace/base_kernel_stats.md
rocprofv3 --kernel-trace -d $O/trace/after -o t -- python $R/bench.py $SHORT --klt-after-intracam 1 > $O/trace/after_line.json 2> $O/trace/after.err
python $R/tools/rocpd_summary.py kernels $O/trace/after/t_results.db --last-frames 300 > $O/trace/after_kernel_stats.md
rm -rf $O/trace/base $O/trace/after
ls -la $O $O/drift $O/ab $O/trace
