#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/merge_check
mkdir -p $O
cd $R
python -m pytest tests/test_poseupdate_gpu.py tests/test_register_decide_gpu.py tests/test_cxx_dropin_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
bash tools/r06_merge_print.sh
