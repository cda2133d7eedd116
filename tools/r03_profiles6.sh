#!/bin/bash
# closing trace of round 3: per-queue durations and gaps of the key-frame solves in the headline loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p6; mkdir -p $O
cd /tmp
rm -rf /tmp/kt && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $R/bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg > $O/traced.log 2> /tmp/kt.err; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/ba_gaps.py $DB > $O/ba_in_loop_durations_and_gaps.txt 2>&1
python $R/tools/key_interval.py $DB > $O/key_interval.txt 2>&1
grep '^{"metric"' $O/traced.log | tail -1 > $O/traced_bench_line.json
cat $O/ba_in_loop_durations_and_gaps.txt | head -40
head -30 $O/key_interval.txt
