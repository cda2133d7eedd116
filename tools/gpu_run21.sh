cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02r; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$?"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench(default) rc=$?"
cd /tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/kernel_stats.md | head -14
python $GRAFT_REPO_ROOT/tools/rocpd_by_grid.py $DB $GRAFT_REPO_ROOT/$O/by_grid.md > /dev/null 2>&1
