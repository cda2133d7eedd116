#!/bin/bash
# round 6's closing measurements in one gpurun call: the GPU suite, smoke, the headline command traced (kernel table + a few frames' kernels per
# stream), the bench lines (default and the driver's command) -> gpurun_out/r06/; tools/r06_collect.py copies them into profiles/r06_*
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tools/r06_trace.sh r06/trace > $O/trace.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver2.json 2> $O/bench_driver2.err
tail -3 $O/gpu_suite.log; tail -2 $O/smoke.log
python - <<'PY'
import json
for n in ("bench_default", "bench_driver", "bench_driver2"):
    try:
        d = json.loads(open(f"gpurun_out/r06/{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["value_source"], d["roofline"]["frac"], d["cpu_baseline"] and d["cpu_baseline"]["value"])
    except Exception as e:
        print(n, "failed", e)
PY
