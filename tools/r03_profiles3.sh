#!/bin/bash
# after the tracker's instruction trimming: SQ pass + kernel stats of the 8-camera tracker, and of cfg5's KLT stage
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p3; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmc_sq; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d /tmp/pmc_sq -o klt -- python $R/tools/pmc_klt.py > /tmp/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
python $R/tools/rocpd_pmc.py $(find /tmp/pmc_sq -name "*.db" | head -1) $O/klt_pmc_SQ.md | grep -E "k_track_rows" | head -8
rm -rf /tmp/k8; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k8 -o klt -- python $R/tools/pmc_klt.py > /tmp/k8.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/k8 -name "*.db" | head -1) $O/klt_8cam_kernel_stats.md | head -6
export GROUP_CAM_CFG5=1 PMC_CAMS=4
rm -rf /tmp/p5_sq; timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d /tmp/p5_sq -o klt -- python $R/tools/pmc_klt.py > /tmp/p5_sq.log 2>&1; echo "cfg5 pmc sq rc=$?"
python $R/tools/rocpd_pmc.py $(find /tmp/p5_sq -name "*.db" | head -1) $O/cfg5_klt_pmc_SQ.md | grep -E "k_track" | head -4
rm -rf /tmp/k5; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/k5 -o klt -- python $R/tools/pmc_klt.py > /tmp/k5.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/k5 -name "*.db" | head -1) $O/cfg5_klt_kernel_stats.md | head -6
