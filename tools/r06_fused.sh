#!/bin/bash
# the registration's fused launches + the scaled forward differences of k_intracam: the GPU tests that pin them, then the C++ loop's value
# with the fused launches on / off on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/fused
mkdir -p $O
cd $R
python -m pytest tests/test_pose_ba_gpu.py tests/test_cxx_dropin_gpu.py tests/test_poseupdate_gpu.py tests/test_register_decide_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
SHORT="--no-cpu-baseline --no-secondary --no-upload-leg --live-pmc 0"
: > $O/ab.txt
for fused in 1 0 1 0; do
  for st in "300 30" "20 5"; do
    set -- $st
    COSLAM_FUSED_ROUNDS=$fused python bench.py $SHORT --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused launches %s steps %4d: value %8.1f frames/s' % ('$fused', d['steps'], d['value']))" >> $O/ab.txt
  done
done
cat $O/ab.txt
