cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$?"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench(default) rc=$?"
python tools/pg_time.py > $O/posegraph_time.txt 2>&1; tail -2 $O/posegraph_time.txt
for m in joint intercam; do python3 bench.py --no-cpu-baseline --only-solve $m 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('only-solve $m:', round(j['value'],1), 'frames/s')"; done | tee $O/only_solve.txt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o klt -- python $GRAFT_REPO_ROOT/tools/pmc_klt.py > /tmp/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  DB=$(find /tmp/pmc_$c -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB $GRAFT_REPO_ROOT/$O/klt_pmc_$c.md | head -5
done
rm -rf /tmp/pmc_sq; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_sq -o klt -- python $GRAFT_REPO_ROOT/tools/pmc_klt.py > /tmp/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
DB=$(find /tmp/pmc_sq -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB $GRAFT_REPO_ROOT/$O/klt_pmc_SQ.md | grep -E "k_track_rows" | head -10
rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/kernel_stats.md | head -16
python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB > $GRAFT_REPO_ROOT/$O/ba_gaps.txt; cat $GRAFT_REPO_ROOT/$O/ba_gaps.txt
