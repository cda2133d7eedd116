#!/bin/bash
# quick check: the loop-level equality tests, then the C++ loop's value (bench, short legs only) twice at 300 and 20 steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/quick
mkdir -p $O
cd $R
python -m pytest tests/test_cxx_dropin_gpu.py tests/test_poseupdate_gpu.py tests/test_register_decide_gpu.py tests/test_register_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
SHORT="--no-cpu-baseline --no-secondary --no-upload-leg --live-pmc 0"
: > $O/ab.txt
for rep in 1 2; do
  for st in "300 30" "20 5"; do
    set -- $st
    python bench.py $SHORT --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps %4d: value %8.1f frames/s' % (d['steps'], d['value']))" >> $O/ab.txt
  done
done
cat $O/ab.txt
