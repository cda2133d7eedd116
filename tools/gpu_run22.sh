cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
nproc; 
b() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value'],1), round(j['ms_per_step'],4), 'host', round(j['config']['host_enqueue_ms_per_step'],3))"; }
b fresh
timeout 600 python -m pytest tests/test_comm_gpu.py tests/test_cxx_dropin_gpu.py -q -m gpu 2>&1 | tail -1
ps aux --sort=-%cpu | head -5
b after_comm_cxx
timeout 900 python -m pytest tests/test_klt_gpu.py tests/test_pose_ba_gpu.py -q -m gpu 2>&1 | tail -1
b after_klt_ba
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b after_smoke
b again
