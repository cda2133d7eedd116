#!/bin/bash
# A/B of the headline loop with legs switched off (diagnostic: which leg costs what, untraced).  Each line: flags -> frames/s, ms/step
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06ab.txt}
SHORT="--no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --live-pmc 0"
: > $O
run() {
  for rep in 1 2; do
    python $R/bench.py $SHORT "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %8.1f frames/s  %.4f ms/step  steps %d' % ('$*', d['value'], d['ms_per_step'], d['steps']))" >> $O
  done
}
run --steps 300 --warmup 30
run --steps 300 --warmup 30 --no-ncc
run --steps 300 --warmup 30 --merge-every 0
run --steps 300 --warmup 30 --no-ncc --merge-every 0
run --steps 300 --warmup 30 --no-ncc --merge-every 0 --key-every 0
run --steps 300 --warmup 30 --no-ncc --merge-every 0 --no-decide
run --steps 300 --warmup 30 --no-ncc --merge-every 0 --no-decide --no-mergability
run --steps 300 --warmup 30 --no-ncc --merge-every 0 --no-decide --no-mergability --no-register
run --steps 300 --warmup 30 --no-ncc --merge-every 0 --no-decide --no-mergability --no-register --no-classify
run --steps 20 --warmup 5
cat $O
