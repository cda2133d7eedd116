cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
timeout 600 python -m pytest tests/test_klt_gpu.py -x -q -m gpu -k "pyramid or prefetch or group" 2>&1 | tail -3
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o klt -- python $GRAFT_REPO_ROOT/tools/pmc_klt.py > /tmp/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  DB=$(find /tmp/pmc_$c -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB $GRAFT_REPO_ROOT/$O/klt_pmc_$c.md | head -12
done
rm -rf /tmp/pmc_sq; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_sq -o klt -- python $GRAFT_REPO_ROOT/tools/pmc_klt.py > /tmp/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
DB=$(find /tmp/pmc_sq -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB $GRAFT_REPO_ROOT/$O/klt_pmc_SQ.md | grep -E "k_track_rows|k_tail" | head -30
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcb_$c -o ba -- python $GRAFT_REPO_ROOT/tools/pmc_ba.py > /tmp/pmcb_$c.log 2>&1; echo "pmc ba $c rc=$?"; tail -2 /tmp/pmcb_$c.log
  DB=$(find /tmp/pmcb_$c -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB $GRAFT_REPO_ROOT/$O/ba_pmc_$c.md | head -14
done
rm -rf /tmp/pmcb_sq; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d /tmp/pmcb_sq -o ba -- python $GRAFT_REPO_ROOT/tools/pmc_ba.py > /tmp/pmcb_sq.log 2>&1; echo "pmc ba sq rc=$?"
DB=$(find /tmp/pmcb_sq -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB $GRAFT_REPO_ROOT/$O/ba_pmc_SQ.md > /dev/null
rm -rf /tmp/cfg5; PMC_CFG5=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cfg5 -o ba -- python $GRAFT_REPO_ROOT/tools/pmc_ba.py > /tmp/cfg5.log 2>&1; echo "cfg5 rc=$?"; tail -2 /tmp/cfg5.log
DB=$(find /tmp/cfg5 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/$O/ba_cfg5_kernel_stats.md | head -16
