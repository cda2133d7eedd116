cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s10; mkdir -p $O
timeout 900 python -m pytest tests/test_klt_gpu.py tests/test_bench_contract_gpu.py -x -q -m gpu 2>&1 | tail -12 | tee $O/pytest.txt
for rep in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_$rep.json 2> $O/bench_driver_$rep.err; echo "bench rc=$?"
python - $O/bench_driver_$rep.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j['value'], j['config']['with_upload'], j['config']['secondary_cfg2'])
PY
done
python3 bench.py --no-cpu-baseline --no-secondary > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - $O/bench_default.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j['value'], j['config']['with_upload'])
PY
