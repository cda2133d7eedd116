# cfg5 bundle adjustment under rocprofv3: kernel table + the matrix-core counters of k_syrk_mfma / k_cholflow (separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; mkdir -p $O
rm -rf /tmp/kt5 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o b -- python $R/tools/ba_cfg5_time.py 1 > /tmp/kt5.log 2>&1; echo "stats rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/kt5 -name "*.db" | head -1) $O/ba_cfg5_kernel_stats.md | head -14
grep SYRK /tmp/kt5.log
python $R/tools/ba_cfg5_time.py 1 0 2>&1 | grep SYRK
rm -rf /tmp/pm5 && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace -d /tmp/pm5 -o b -- python $R/tools/ba_cfg5_time.py 1 > /tmp/pm5.log 2>&1; echo "pmc rc=$?"
python $R/tools/rocpd_pmc.py $(find /tmp/pm5 -name "*.db" | head -1) $O/ba_cfg5_pmc_mfma.md | grep -E "k_syrk_mfma|k_cholflow " | head -14
tail -3 /tmp/pm5.log
