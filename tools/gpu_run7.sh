cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
rocm-smi --showid 2>/dev/null | head -5; python -c "import torch; print('gpus', torch.cuda.device_count())"
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_handback_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_new.log
# N=2 code path of the bench on ONE GPU: gloo instead of RCCL (test hook), both ranks on device 0
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 5 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "bench n2 rc=$?"; tail -3 $O/bench_n2_gloo.err; python -c "
import json; d=json.load(open('$O/bench_n2_gloo.json')); print('N=2 (gloo hook, one GPU):', d['value'], d['ms_per_step'], d['config']['collectives'], d['config']['pose_ok'])"
