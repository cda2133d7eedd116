#!/bin/bash
# round-3 profile set: headline kernel trace (+ key interval / BA gaps), the 8-camera tracker's PMC passes, cfg5's KLT stage
# (4 x 1920 x 1080 x 5000 slots): timing, kernel stats, PMC passes.  Outputs under gpurun_out/r03p (copied into profiles/r03_*).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
# 1. headline, the driver's command, under the kernel trace
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $R/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 100 --warmup 10 > $O/headline_traced_bench.json 2> /tmp/kt.err; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $O/headline_kernel_stats.md | head -12
python $R/tools/ba_gaps.py $DB > $O/ba_in_loop_durations_and_gaps.txt 2>&1
python $R/tools/key_interval.py $DB > $O/key_interval.txt 2>&1
# 2. the 8-camera tracker: PMC passes (separate), all cameras in one launch
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o klt -- python $R/tools/pmc_klt.py > /tmp/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc_$c -name "*.db" | head -1) $O/klt_pmc_$c.md | head -4
done
rm -rf /tmp/pmc_sq; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_sq -o klt -- python $R/tools/pmc_klt.py > /tmp/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
python $R/tools/rocpd_pmc.py $(find /tmp/pmc_sq -name "*.db" | head -1) $O/klt_pmc_SQ.md | grep -E "k_track_rows" | head -10
rm -rf /tmp/k8; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k8 -o klt -- python $R/tools/pmc_klt.py > /tmp/k8.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/k8 -name "*.db" | head -1) $O/klt_8cam_kernel_stats.md | head -8
# 3. cfg5's KLT stage
export GROUP_CAM_CFG5=1 PMC_CAMS=4
python $R/tools/group_cam.py > $O/cfg5_klt_stage_timing.txt 2>&1; cat $O/cfg5_klt_stage_timing.txt
rm -rf /tmp/k5; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/k5 -o klt -- python $R/tools/pmc_klt.py > /tmp/k5.log 2>&1; echo "cfg5 kt rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/k5 -name "*.db" | head -1) $O/cfg5_klt_kernel_stats.md | head -10
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p5_$c; timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/p5_$c -o klt -- python $R/tools/pmc_klt.py > /tmp/p5_$c.log 2>&1; echo "cfg5 pmc $c rc=$?"
  python $R/tools/rocpd_pmc.py $(find /tmp/p5_$c -name "*.db" | head -1) $O/cfg5_klt_pmc_$c.md | head -4
done
rm -rf /tmp/p5_sq; timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d /tmp/p5_sq -o klt -- python $R/tools/pmc_klt.py > /tmp/p5_sq.log 2>&1; echo "cfg5 pmc sq rc=$?"
python $R/tools/rocpd_pmc.py $(find /tmp/p5_sq -name "*.db" | head -1) $O/cfg5_klt_pmc_SQ.md | grep -E "k_track" | head -10
unset GROUP_CAM_CFG5 PMC_CAMS
