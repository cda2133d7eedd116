#!/bin/bash
mkdir -p gpurun_out/r03_12
timeout 900 python -m pytest tests/test_pose_ba_gpu.py tests/test_cxx_dropin_gpu.py -x -q -k "window or dropin or packed or async" > gpurun_out/r03_12/pytest.txt 2>&1; tail -5 gpurun_out/r03_12/pytest.txt
for k in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r03_12/bench_$k.json 2> gpurun_out/r03_12/bench_$k.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03_12/bench_$k.json').read().strip().splitlines()[-1])
    c=d['config']; print('$k', round(d['value'],1), 'cxx', c['cxx_frame_loop'].get('frames_per_s'), 'upload', c['with_upload']['frames_per_s'] if c.get('with_upload') else None, c['joint_ba_last'])
except Exception as e:
    print('$k FAILED', e); print(open('gpurun_out/r03_12/bench_$k.err').read()[-1500:])
PY
done
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > gpurun_out/r03_12/bench_300.json 2> gpurun_out/r03_12/bench_300.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_12/bench_300.json').read().strip().splitlines()[-1]); c=d['config']
print('300 steps', round(d['value'],1), 'cxx', c['cxx_frame_loop'].get('frames_per_s'), c['joint_ba_last'])"
cd /tmp
rm -rf /tmp/kt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg --steps 100 --warmup 10 > /tmp/kt.log 2>&1; echo "kt rc=$?"
DB=$(find /tmp/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB $GRAFT_REPO_ROOT/gpurun_out/r03_12/kernel_stats.md | head -16
python $GRAFT_REPO_ROOT/tools/ba_gaps.py $DB > $GRAFT_REPO_ROOT/gpurun_out/r03_12/ba_gaps.txt 2>&1; tail -12 $GRAFT_REPO_ROOT/gpurun_out/r03_12/ba_gaps.txt
