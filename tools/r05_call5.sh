#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -6
COSLAM_MERGE_DEBUG=1 timeout 300 python tools/r05_drift.py --variant full --frames 420 --every 100 --out $O/dbg.jsonl > $O/dbg.log 2>&1
grep k_decide_merge $O/dbg.log | tail -4
SHORT="--no-cpu-baseline --no-secondary --no-upload-leg"
timeout 240 python bench.py $SHORT > $O/base.json 2> $O/base.err; python - <<PY
import json
d = json.loads(open("$O/base.json").read().strip().splitlines()[-1]); print("base", round(d["value"], 1), "cxx", d["config"]["cxx_frame_loop"].get("frames_per_s"))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py $SHORT --no-cxx-loop > $O/base_line.json 2> $O/trace.err
python $R/tools/rocpd_summary.py kernels $O/trace/t_results.db --last-frames 300 > $O/base_kernel_stats.md
rm -rf $O/trace
head -34 $O/base_kernel_stats.md
