#!/bin/bash
O=gpurun_out/r03_26; mkdir -p $O
timeout 900 python -m pytest tests/test_poseupdate_gpu.py tests/test_pose_ba_gpu.py -x -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
