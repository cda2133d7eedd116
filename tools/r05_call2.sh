#!/bin/bash
# GPU call 2 of round 5: the suite with the running whole-track mergability verdict + the list-driven registration, the drift run with
# the registration now attaching, the headline loop (Python + C++) and its kernel trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05e
mkdir -p $O/drift $O/ab $O/trace
cd $R
( time python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
for v in full no_ncc; do
  timeout 200 python tools/r05_drift.py --variant $v --frames 1500 --count-attach --time-intracam --out $O/drift/$v.jsonl > $O/drift/$v.log 2>&1 || echo "drift $v rc=$?"
  tail -3 $O/drift/$v.log
done
SHORT="--no-cpu-baseline --no-secondary --no-upload-leg"
ab() { name=$1; shift; timeout 240 python bench.py $SHORT "$@" > $O/ab/$name.json 2> $O/ab/$name.err || echo "ab $name rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("$O/ab/$name.json").read().strip().splitlines()[-1])
    c = d["config"]
    print("$name", round(d["value"], 1), "frames/s", "pose_err", round(c["pose_translation_error_vs_truth"], 4), "cxx", (c.get("cxx_frame_loop") or {}).get("frames_per_s"))
    print("   reg", json.dumps(c.get("register_candidates_last_frame"))[:900])
    print("   dec", json.dumps({k: v for k, v in (c.get("register_decision") or {}).items() if k != "what" and k != "merge"}))
except Exception as e:
    print("$name", "FAILED", e, open("$O/ab/$name.err").read()[-600:])
PY
}
ab base1

ab base2 --no-cxx-loop
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace/base -o t -- python $R/bench.py $SHORT --no-cxx-loop > $O/trace/base_line.json 2> $O/trace/base.err
python $R/tools/rocpd_summary.py kernels $O/trace/base/t_results.db --last-frames 300 > $O/trace/base_kernel_stats.md
rm -rf $O/trace/base
head -40 $O/trace/base_kernel_stats.md
