# rocprofv3 passes behind profiles/r02_*: KLT stage of the 8-camera group (kernel stats), BA kernels (FETCH / WRITE / SQ in
# separate passes, kernel stats), cfg5-shaped BA (kernel stats).  Run on the GPU box: bash tools/profile_run.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o klt -- python $R/tools/pmc_klt.py > /tmp/kt.log 2>&1; echo "klt stats rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) $O/klt_8cam_kernel_stats.md | head -6
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pb_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pb_$c -o ba -- python $R/tools/pmc_ba.py > /tmp/pb_$c.log 2>&1; echo "ba pmc $c rc=$?"
  python $R/tools/rocpd_pmc.py $(find /tmp/pb_$c -name "*.db" | head -1) $O/ba_pmc_$c.md | head -7
done
rm -rf /tmp/pb_sq; timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d /tmp/pb_sq -o ba -- python $R/tools/pmc_ba.py > /tmp/pb_sq.log 2>&1; echo "ba sq rc=$?"
python $R/tools/rocpd_pmc.py $(find /tmp/pb_sq -name "*.db" | head -1) $O/ba_pmc_SQ.md > /dev/null
rm -rf /tmp/kb; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kb -o ba -- python $R/tools/pmc_ba.py > /tmp/kb.log 2>&1; echo "ba stats rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/kb -name "*.db" | head -1) $O/ba_kernel_stats.md | head -12
rm -rf /tmp/c5; PMC_CFG5=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/c5 -o ba -- python $R/tools/pmc_ba.py > /tmp/c5.log 2>&1; echo "cfg5 rc=$?"
python $R/tools/rocpd_summary.py $(find /tmp/c5 -name "*.db" | head -1) $O/ba_cfg5_kernel_stats.md | head -8
