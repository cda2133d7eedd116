cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/timeline.py $DB 1200 0.5 > $GRAFT_REPO_ROOT/gpurun_out/r02n/timeline.txt; wc -l $GRAFT_REPO_ROOT/gpurun_out/r02n/timeline.txt
