cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
timeout 1200 python -m pytest tests/test_pose_ba_gpu.py tests/test_sliced_ba_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest_ba.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_ba.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['config']['joint_ba_last'], d['config']['intercam_last'], d['config']['host_enqueue_ms_per_step'])"; tail -3 $O/bench.err
COSLAM_BA_LEGACY_SOLVE=1 timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_legacy.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_legacy.json')); print('legacy solver:', d['value'], d['ms_per_step'])"
for ch in 2 8; do COSLAM_BA_CHUNK=$ch timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_chunk$ch.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_chunk$ch.json')); print('chunk $ch:', d['value'], d['ms_per_step'])"; done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/prof1.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof1 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB $O/kernel_stats.md | head -40; fi
