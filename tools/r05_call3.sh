#!/bin/bash
# GPU call: the two readings of Const::PIXEL_ERR_VAR side by side (drift curves + headline loop), then the suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05f
mkdir -p $O/drift $O/ab $O/trace
cd $R
for rd in variance std; do
  for v in full no_ncc; do
    timeout 300 python tools/r05_drift.py --variant $v --frames 1500 --count-attach --time-intracam --pixel-err-reading $rd --out $O/drift/${v}_$rd.jsonl > $O/drift/${v}_$rd.log 2>&1 || echo "drift $v $rd rc=$?"
  done
done
SHORT="--no-cpu-baseline --no-secondary --no-upload-leg"
ab() { name=$1; shift; timeout 240 python bench.py $SHORT "$@" > $O/ab/$name.json 2> $O/ab/$name.err || echo "ab $name rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("$O/ab/$name.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("$name", round(d["value"], 1), "frames/s", "rig", json.dumps(c.get("rig_error_vs_truth"))[:300], "cxx", (c.get("cxx_frame_loop") or {}).get("frames_per_s"))
    print("   reg", json.dumps({k: v for k, v in (c.get("register_candidates_last_frame") or {}).items() if k not in ("running_verdict", "active_search")}))
    print("   dec", json.dumps({k: v for k, v in (c.get("register_decision") or {}).items() if k != "what"})[:700])
    print("   map", json.dumps(c.get("map_error_vs_truth"))[:400], "false", c["pose_update"]["map_points_classify"]["map_points_false"], "BA", c["joint_ba_problem"]["points"], c["joint_ba_last"])
except Exception as e:
    print("$name", "FAILED", e, open("$O/ab/$name.err").read()[-600:])
PY
}
ab var1 --pixel-err-reading variance
ab std1 --no-cxx-loop --pixel-err-reading std
ab var2 --no-cxx-loop --pixel-err-reading variance
( time python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace/base -o t -- python $R/bench.py $SHORT --no-cxx-loop > $O/trace/base_line.json 2> $O/trace/base.err
python $R/tools/rocpd_summary.py kernels $O/trace/base/t_results.db --last-frames 300 > $O/trace/base_kernel_stats.md
rm -rf $O/trace/base
head -36 $O/trace/base_kernel_stats.md
