cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pose_ba_gpu.py -x -q -m gpu 2>&1 | tail -2
for sp in 1 0 1 0; do
COSLAM_BA_SPECULATE=$sp python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec $sp K=20', round(j['value'],1), round(j['ms_per_step'],4), j['config']['joint_ba_last']['lm_steps'], j['config']['intercam_last']['lm_steps'])"
COSLAM_BA_SPECULATE=$sp python3 bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec $sp K=300', round(j['value'],1), round(j['ms_per_step'],4))"
done
