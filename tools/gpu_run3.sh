cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
timeout 900 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; tail -5 $O/bench.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --key-every 0 > $O/bench_nokey.json 2>> $O/bench.err; cat $O/bench_nokey.json | python -c "import sys,json; d=json.load(sys.stdin); print('no key-frame solves:', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --key-every 0 --no-pose > $O/bench_kltonly.json 2>> $O/bench.err; cat $O/bench_kltonly.json | python -c "import sys,json; d=json.load(sys.stdin); print('klt only:', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --serial > $O/bench_serial.json 2>> $O/bench.err; cat $O/bench_serial.json | python -c "import sys,json; d=json.load(sys.stdin); print('serial:', d['value'], d['ms_per_step'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/prof1.log 2>&1; echo "rocprof rc=$?"; tail -3 /tmp/prof1.log
cd $GRAFT_REPO_ROOT
find /tmp/prof1 -name "*.db" | head; DB=$(find /tmp/prof1 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB $O/kernel_stats.md | head -50; fi
find /tmp/prof1 -name "*stats*" | head
