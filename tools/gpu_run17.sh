cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
BENCH_DIST_BACKEND=gloo BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 40 --warmup 5 > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc=$?"; tail -c 600 $O/bench_n2.err; grep '"metric"' $O/bench_n2.json | head -c 600
