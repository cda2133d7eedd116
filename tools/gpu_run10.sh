cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
timeout 900 python -m pytest tests/test_klt_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/group_cam.py quick 2>&1 | tail -8
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o klt -- python $GRAFT_REPO_ROOT/tools/pmc_klt.py > /tmp/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  DB=$(find /tmp/pmc_$c -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB $GRAFT_REPO_ROOT/$O/klt_pmc_$c.md | head -9
done
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o klt -- python $GRAFT_REPO_ROOT/tools/pmc_klt.py > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/klt_kernel_stats.md | head -12
