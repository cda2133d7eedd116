#!/bin/bash
# the closing validation of a round on the GPU box: the full GPU suite, smoke(), the default bench command and the driver's, each timed
mkdir -p gpurun_out/final
o=gpurun_out/final
(time python -m pytest tests -m gpu -q) > $o/pytest_gpu.log 2>&1; tail -4 $o/pytest_gpu.log
(time python -c "import __graft_entry__ as g; g.smoke()") > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $o/smoke.log
(time python bench.py) > $o/bench_default.json 2> $o/bench_default.err; tail -4 $o/bench_default.err
(time python bench.py --gpus 1 --steps 20 --warmup 5) > $o/bench_driver.json 2> $o/bench_driver.err; tail -4 $o/bench_driver.err
python - <<PY
import json
for f in ("$o/bench_default.json", "$o/bench_driver.json"):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(j["value"], 1), "frames/s", j["roofline"]["frac"], j["roofline"]["traffic"], j["cpu_baseline"]["value"])
PY
