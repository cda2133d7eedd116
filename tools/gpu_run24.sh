cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB
