#!/bin/bash
# the classification over feature references in the loop: the GPU tests that pin it, then the headline (C++ loop's value) with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/classify_refs
mkdir -p $O
cd $R
python -m pytest tests/test_poseupdate_gpu.py tests/test_cxx_dropin_gpu.py tests/test_keyframe_drives_gpu.py tests/test_bench_contract_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
SHORT="--no-cpu-baseline --no-secondary --no-upload-leg --live-pmc 0"
: > $O/ab.txt
for plain in 0 1 0 1; do
  for st in "300 30" "20 5"; do
    set -- $st
    if [ $plain = 1 ]; then export COSLAM_CLASSIFY_PLAIN=1; else unset COSLAM_CLASSIFY_PLAIN; fi
    python bench.py $SHORT --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config'].get('cxx_frame_loop') or {}
print('classify over references %s steps %4d: value %8.1f frames/s (python loop %s) map false %s dynamic %s' % ('off' if '$plain'=='1' else 'on ', d['steps'], d['value'], (d['config'].get('python_frame_loop') or {}).get('value'), c.get('map_points_false'), c.get('map_points_dynamic')))" >> $O/ab.txt
  done
done
cat $O/ab.txt
