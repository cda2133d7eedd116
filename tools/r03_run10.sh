#!/bin/bash
mkdir -p gpurun_out/r03_10
./oracle/_ref/ref_ba_dropin_test > gpurun_out/r03_10/dropin.txt 2>&1; echo "dropin rc=$?"; tail -4 gpurun_out/r03_10/dropin.txt
timeout 900 python -m pytest tests/test_poseupdate_gpu.py tests/test_cxx_dropin_gpu.py tests/test_handback_gpu.py -x -q > gpurun_out/r03_10/pytest.txt 2>&1; tail -15 gpurun_out/r03_10/pytest.txt
