#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
COSLAM_MERGE_DEBUG=1 timeout 300 python tools/r05_drift.py --variant full --frames 420 --every 100 --out $O/dbg.jsonl > $O/dbg.log 2>&1
grep k_decide_merge $O/dbg.log | tail -3
