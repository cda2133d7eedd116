#!/bin/bash
# BASELINE.json configs[4]'s KLT stage (4 cameras 1920 x 1080 x 5000 slots as one camera group): kernel trace + the three counter passes of
# GROUP_CAM_CFG5=1 PMC_CAMS=4 tools/pmc_klt.py -> gpurun_out/r06/cfg5/; tools/r06_cfg5_collect.py turns them into profiles/r06_cfg5_klt_pmc.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/cfg5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GROUP_CAM_CFG5=1 PMC_CAMS=4
rocprofv3 --kernel-trace -d $O/kt -o p -- python $R/tools/pmc_klt.py > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py kernels $O/kt/p_results.db > $O/cfg5_klt_kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p -- python $R/tools/pmc_klt.py > $O/$c.log 2>&1
  python $R/tools/rocpd_summary.py counters $O/$c/p_results.db > $O/cfg5_klt_pmc_$c.md
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace -d $O/SQ -o p -- python $R/tools/pmc_klt.py > $O/SQ.log 2>&1
python $R/tools/rocpd_summary.py counters $O/SQ/p_results.db > $O/cfg5_klt_pmc_SQ.md
rm -rf $O/kt $O/FETCH_SIZE $O/WRITE_SIZE $O/SQ
grep "track_rows" $O/*.md | cut -c1-200
