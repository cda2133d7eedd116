cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_s15; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.txt
timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b.json 2> $O/b.err; echo rc=$?
python - $O/b.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j['config']
    print("value", round(j['value'],1), "upload", round(c['with_upload']['frames_per_s'],1), "cxx", c['cxx_frame_loop'])
except Exception as e:
    print('FAILED',e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
