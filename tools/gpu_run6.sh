cd $GRAFT_REPO_ROOT
COSLAM_HIP_LIB=$GRAFT_REPO_ROOT/coslam_amd/lib/libcoslam_hip_probe.so timeout 300 python -m pytest tests/test_pose_ba_gpu.py -x -q -m gpu -k "joint_local or inter_camera_pose" -s 2>&1 | grep -E "k_solve_blocked|passed|failed" | head
timeout 600 python -m pytest tests/test_pose_ba_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.load(sys.stdin); print('bench:', d['value'], d['ms_per_step'])"
