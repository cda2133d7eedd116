set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests/test_klt_gpu.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r02a/pytest_klt.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a/pytest_klt.log
tail -30 gpurun_out/r02a/pytest_klt.log
timeout 600 python tools/group_cam.py > gpurun_out/r02a/group_cam.log 2>&1
echo "group_cam rc=$?" >> gpurun_out/r02a/group_cam.log
cat gpurun_out/r02a/group_cam.log
