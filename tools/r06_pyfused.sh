#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/pyfused
mkdir -p $O
cd $R
python -m pytest tests/test_keyframe_drives_gpu.py tests/test_cxx_dropin_gpu.py tests/test_register_decide_gpu.py tests/test_register_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
SHORT="--no-cpu-baseline --no-secondary --no-upload-leg --live-pmc 0"
python bench.py $SHORT --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'python loop', d['config']['python_frame_loop']['frames_per_s'])"
python bench.py $SHORT --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'python loop', d['config']['python_frame_loop']['frames_per_s'])"
