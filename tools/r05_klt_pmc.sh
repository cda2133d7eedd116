#!/bin/bash
# the KLT stage's three counter passes alone (section 2 of tools/r05_profiles.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/klt_$c -o p -- python $R/tools/pmc_klt.py > $O/klt_$c.log 2>&1
  python $R/tools/rocpd_summary.py counters $O/klt_$c/p_results.db > $O/klt_pmc_$c.md
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $O/klt_SQ -o p -- python $R/tools/pmc_klt.py > $O/klt_SQ.log 2>&1
python $R/tools/rocpd_summary.py counters $O/klt_SQ/p_results.db > $O/klt_pmc_SQ.md
rm -rf $O/klt_FETCH_SIZE $O/klt_WRITE_SIZE $O/klt_SQ
head -4 $O/klt_pmc_FETCH_SIZE.md
