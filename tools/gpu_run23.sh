cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_sliced_ba_gpu.py tests/test_configs_gpu.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
python tools/ba_time.py 2>&1 | tail -2
for k in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20', round(j['value'],1), round(j['ms_per_step'],4))"; done
python3 bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=300', round(j['value'],1), round(j['ms_per_step'],4))"
