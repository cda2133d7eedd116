cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s11; mkdir -p $O
timeout 600 python -m pytest tests/test_klt_gpu.py -x -q -m gpu -k staged 2>&1 | tail -3
run() { n=$1; shift
timeout 300 python3 bench.py --no-cpu-baseline --no-secondary "$@" > $O/$n.json 2> $O/$n.err
python - $O/$n.json $n <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); u=j['config']['with_upload']
    print(f"{sys.argv[2]:24s} value {j['value']:8.1f}  upload {u['frames_per_s']:8.1f} ratio {u['ratio_to_value']:.3f}")
except Exception as e:
    print(sys.argv[2],'FAILED',e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
run d300
run d20 --steps 20 --warmup 5
BENCH_UPLOAD_MODE=copyonly run copyonly300
BENCH_UPLOAD_MODE=copyonly run copyonly20 --steps 20 --warmup 5
HSA_ENABLE_SDMA=0 run nosdma300
