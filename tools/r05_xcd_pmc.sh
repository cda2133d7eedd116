#!/bin/bash
# FETCH_SIZE of the tracker launch under the three placements / margins (tools/pmc_klt.py), one rocprofv3 --pmc pass each
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05x/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
M1=$R/coslam_amd/lib/libcoslam_hip_margin1.so
run() {  # name lib xcd counter
  COSLAM_HIP_LIB=$2 KLT_XCD=$3 rocprofv3 --pmc $4 --kernel-trace -d $O/$1_$4 -o p -- python $R/tools/pmc_klt.py > $O/$1_$4.log 2>&1
  python $R/tools/rocpd_summary.py counters $O/$1_$4/p_results.db > $O/$1_pmc_$4.md
  rm -rf $O/$1_$4
  grep -i "track_rows" $O/$1_pmc_$4.md | head -2
}
run base "" 0 FETCH_SIZE
run base_xcd "" 1 FETCH_SIZE
run m1 $M1 0 FETCH_SIZE
run m1_xcd $M1 1 FETCH_SIZE
run m1_xcd $M1 1 WRITE_SIZE
