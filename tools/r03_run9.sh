cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s9; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
python - $O/bench_driver_cmd.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j['value'], json.dumps(j['roofline'])[:1200])
print(j['cpu_baseline'])
print(j['config']['secondary_cfg2'], j['config']['secondary_cfg5_ba'])
PY
