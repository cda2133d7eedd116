cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_s17; mkdir -p $O
timeout 600 python3 bench.py --no-cpu-baseline --no-cxx-loop --no-upload-leg --steps 20 --warmup 5 > $O/b.json 2> $O/b.err; echo rc=$?
python - $O/b.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j['config']
    print("value", round(j['value'],1)); print(c['secondary_cfg5_klt']); print(c['secondary_cfg5_ba'])
except Exception as e:
    print('FAILED',e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
