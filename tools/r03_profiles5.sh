#!/bin/bash
# SQ counters of the pose stream's kernels in the headline loop (k_handback, k_intracam, k_pose_update, k_register_search,
# k_register_mergability, the NCC run's kernels)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p5; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/ps; timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace -d /tmp/ps -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg > /tmp/ps.log 2>&1; echo "rc=$?"
python $R/tools/rocpd_pmc.py $(find /tmp/ps -name "*.db" | head -1) $O/pose_stream_pmc_SQ.md > /dev/null
grep -E "k_intracam|k_handback|k_pose_update|k_register|k_ncc|k_resize" $O/pose_stream_pmc_SQ.md | grep -E "SQ_INSTS_VALU|SQ_WAVE_CYCLES|SQ_WAVES|SQ_WAIT_ANY" | head -30
