#!/bin/bash
# the packed LM-step kernels stand-alone: kernel stats + FETCH / WRITE passes (the headline's pre-baked joint BA and inter-camera solve)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p4; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/kb; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kb -o ba -- python $R/tools/pmc_ba.py > $O/ba_standalone.log 2>&1; echo "ba stats rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/kb -name "*.db" | head -1) $O/ba_kernel_stats_standalone.md | head -14
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pb_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pb_$c -o ba -- python $R/tools/pmc_ba.py > /tmp/pb_$c.log 2>&1; echo "ba pmc $c rc=$?"
  python $R/tools/rocpd_pmc.py $(find /tmp/pb_$c -name "*.db" | head -1) $O/ba_pmc_$c.md | head -8
done
grep "LM steps" $O/ba_standalone.log
