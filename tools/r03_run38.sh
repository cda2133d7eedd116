#!/bin/bash
O=gpurun_out/r03_38; mkdir -p $O
timeout 600 python -m pytest tests/test_ncc_gpu.py tests/test_cxx_dropin_gpu.py -x -q 2>&1 | tail -3
python tools/ncc_time.py 2>&1 | tail -6
for rep in 1 2; do
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-upload-leg --no-cxx-loop > $O/b_$rep.json 2> $O/b_$rep.err
python -c "
import json
d=json.loads(open('$O/b_$rep.json').read().strip().splitlines()[-1]); c=d['config']; print(round(d['value'],1), c['ncc_matching']['pairs_kept_last_run'])"
done
cd /tmp
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 60 --warmup 10 > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats.md | grep -E "ncc|resize|register"
