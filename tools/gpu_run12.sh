cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
timeout 600 python -m pytest tests/test_handback_gpu.py tests/test_register_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py > $O/bench.json 2> $O/bench.err; echo bench rc=$?; tail -c 400 $O/bench.err
python bench.py --no-register --no-cpu-baseline > $O/bench_noreg.json 2>> $O/bench.err; echo rc=$?
cd /tmp && rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/kernel_stats.md | head -16
